#!/bin/bash
# Throughput cost of the round-6 changes to the 256x256 two-group main loop (B staging units by read phase, phase-1 reads retired before the
# barrier): libgdf_oldboth.so (round-5 form, tools/build_variant.sh oldboth -DGDF_EXP_OLD_BUNITS -DGDF_EXP_OLD_LGKM) against libgdf.so, same box,
# alternating.  The kernel carries SDXL's GEGLU GEMMs (22 % of the step), the VAE's 256 / 512-wide
# convs and every MMDiT GEMM.
D=generic-diffusion-feature_amd
cp $D/libgdf.so /tmp/keep.so
for r in 1 2 3; do
  for v in oldboth product; do
    if [ $v = product ]; then cp /tmp/keep.so $D/libgdf.so; else cp $D/libgdf_$v.so $D/libgdf.so; fi
    echo "== $v run $r"
    python3 bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  sdxl B=16      ', d['value'], 'img/s', d['ms_per_step'], 'ms', d['roofline']['achieved'], 'TF dominant')"
    for dt in auto fp8-mx; do
      python3 bench.py --version flux --flux-dtype $dt --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  flux B=8 $dt', d['value'], 'img/s', d['ms_per_step'], 'ms')"
    done
    python3 tools/bench_vae.py --steps 5 2>/dev/null | grep "^VAE" | cut -c1-110
  done
done
cp /tmp/keep.so $D/libgdf.so
