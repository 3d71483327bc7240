#!/bin/bash
# Every MFMA kernel family of the library beside partners of another kind on a second stream, bit for bit (tools/micro/op_race.py).
# Subjects (thread 0) x partners (thread 1); RACE_ITERS launches each.
IT=${RACE_ITERS:-1500}
PARTNERS="layernorm|groupnorm|attention cross B4|gemm res32 4096x320x320|conv3x3 B4 32x32 320->320"
for subj in "geglu M4096 C320 (auto tile)" "geglu M2048 C640 (K=640)" "geglu M16384 C1280 (auto)" "gemm 4096x960x320 tile 932 (256x320 8-phase)" "gemm 16384x1280x1280 tile 932" \
            "conv3x3 B4 32x32 320->640 (auto: 932)" "conv3x3 B2 64x64 256->256 (826)" "conv3x3 B1 128x128 128->128" "dit gemm 4096x3072x3072 (8256) gelu" \
            "dit gemm 2048x1024x256 (8256) short K" "attention self B2 h10 S4096 D64" "attention joint B1 h24 512+1024 D128" "attention cross B4 h8 Sq1024 Sk77 D40" \
            "gemm res32 4096x320x1280 (auto)" "gemm qkv 4096x960x320 (auto)"; do
  echo "== subject (thread 0): $subj"
  RACE_ITERS=$IT RACE_ONLY="$PARTNERS" RACE_PAIR="$subj" timeout 900 python tools/micro/op_race.py 2>&1 | grep "differing" | cut -c1-150
done
