#!/bin/bash
# Which of the two round-6 changes to the 256x256 two-group main loop removes the concurrent-stream corruption?  (run on the GPU box)
#   libgdf_oldboth.so   = round-5 form: B units = row halves, phase-1 reads retired after the barrier   (-DGDF_EXP_OLD_BUNITS -DGDF_EXP_OLD_LGKM)
#   libgdf_unitsonly.so = ONLY the B units follow the read phases                                  (-DGDF_EXP_OLD_LGKM)
#   libgdf_lgkmonly.so  = ONLY the reads are retired before the barrier                            (-DGDF_EXP_OLD_BUNITS)
#   libgdf.so           = both (the product)
D=generic-diffusion-feature_amd
cp $D/libgdf.so /tmp/product.so
for v in oldboth unitsonly lgkmonly product; do
  if [ $v = product ]; then cp /tmp/product.so $D/libgdf.so; else cp $D/libgdf_$v.so $D/libgdf.so; fi
  echo "== $v"
  RACE_ITERS=${RACE_ITERS:-6000} RACE_ONLY=layernorm RACE_PAIR="geglu M4096 C320 (auto tile)" timeout 500 python tools/micro/op_race.py 2>&1 | tail -1 | cut -c1-200
  GDF_HIP_GRAPH=0 RACE_ITERS=300 timeout 300 python tools/micro/thread_race.py 2>&1 | grep "^thread [01]:"
done
cp /tmp/product.so $D/libgdf.so
