# A/B of padded tick strides x block orders (two-pass and single-pass fusion); see tools/stride_ab.py
for pad in 0 4352 69888; do
  python3 tools/stride_ab.py 0 $pad 2>&1 | grep "^mode"
  for ch in 16 1 4 64; do LSN_FUSE_CHUNK=$ch python3 tools/stride_ab.py 2 $pad 2>&1 | grep "^mode"; done
done
