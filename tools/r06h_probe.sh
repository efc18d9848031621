O=gpurun_out/r06h; mkdir -p $O
V=$PWD/faceoff_amd/csrc/variants
for round in 1 2; do
for v in new st1 st2 st3; do
  echo "== $v (round $round)" >> $O/h64_stagger.txt
  L=$V/lib_$v.so; [ $v = new ] && L=$PWD/faceoff_amd/libfaceoff_hip.so
  FACEOFF_HIP_LIB=$L python tools/bench_bf16.py "conv1_2" 2>&1 | grep -v "amdgpu.ids\|sum" >> $O/h64_stagger.txt
  FACEOFF_HIP_LIB=$L python tools/bench_bf16.py "conv2_1 fwd" 2>&1 | grep -v "amdgpu.ids\|sum" >> $O/h64_stagger.txt
done; done
cat $O/h64_stagger.txt
