#!/bin/bash
# the source split of csrc/gemm.hip (gemm_common.h / gemm_tile.h / gemm_mainloop_ring.h / gemm_mainloop_8phase.h / gemm_epilogue.h)
# against the single-function source it replaces: same box, alternating.  libgdf_prev.so = the pre-split gemm.hip object + the same other objects,
# libgdf_new.so = the split sources (both through tools/build_variant.sh's flags).  Every bench line that the GEMM family carries.
D=generic-diffusion-feature_amd
cp $D/libgdf.so /tmp/keep.so
for r in 1 2; do
  for v in prev new; do
    cp $D/libgdf_$v.so $D/libgdf.so
    echo "== $v run $r"
    python3 bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  sdxl B=16      ', d['value'], 'img/s', d['ms_per_step'], 'ms', d['roofline']['achieved'], 'TF dominant')"
    python3 bench.py --version 1-5 --batch 32 --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  sd1.5 B=32     ', d['value'], 'img/s', d['ms_per_step'], 'ms')"
    for dt in auto bfloat16 fp8-mx bfloat16x2; do
      python3 bench.py --version flux --flux-dtype $dt --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  flux B=8 $dt', d['value'], 'img/s', d['ms_per_step'], 'ms')"
    done
    python3 tools/bench_vae.py --steps 5 2>/dev/null | grep "^VAE" | cut -c1-110
    python3 tools/bench_pixart.py 2>/dev/null | tail -2 | cut -c1-160
  done
done
cp /tmp/keep.so $D/libgdf.so
