#!/bin/bash
cd /root/repo
for sh in 16,256,256,128,0,128,3,2 16,256,256,256,0,128,1,1 16,256,256,128,0,128,3,1; do
  for rep in 1 2; do
  (cd _old_r02 && python tools/conv16_bench.py --shipped --plain --abl 0 --shape $sh 2>&1 | grep -v amdgpu.ids | sed 's/^/old /')
  python tools/conv16_bench.py --shipped --plain --abl 0 --shape $sh 2>&1 | grep -v amdgpu.ids | sed 's/^/new /'
  done
done
