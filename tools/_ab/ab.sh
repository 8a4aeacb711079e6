#!/bin/bash
# same-box A/B of two builds of libasrk.so: tools/_ab/libasrk_old.so vs the in-tree one
L=end-to-end-asr-pytorch_amd/csrc/libasrk.so
cp $L /tmp/new.so
for rep in 1 2; do
for v in old new; do
  if [ $v = old ]; then cp tools/_ab/libasrk_old.so $L; else cp /tmp/new.so $L; fi
  echo "== $v"
  "$@"
done
done
cp /tmp/new.so $L
