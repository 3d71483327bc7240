#!/bin/bash
# same-box A/B of two builds of libgdf.so: tools/ab_lib.sh <command...>   (libgdf_prev.so = A, libgdf.so = B)
D=generic-diffusion-feature_amd
cp $D/libgdf.so /tmp/new.so
for r in 1 2; do
  cp $D/libgdf_prev.so $D/libgdf.so; echo "== A (prev) run $r"; "$@"
  cp /tmp/new.so $D/libgdf.so;       echo "== B (new)  run $r"; "$@"
done
