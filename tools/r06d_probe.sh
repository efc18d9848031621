O=gpurun_out/r06d; mkdir -p $O
V=faceoff_amd/csrc/variants
for round in 1 2; do
for v in old incr incr_a3 incr_p1 incr_a3p1; do
  echo "== $v (round $round)" >> $O/h64_ab.txt
  FACEOFF_HIP_LIB=$PWD/$V/lib_$v.so python tools/bench_bf16.py conv1_2 2>&1 | grep -v amdgpu.ids >> $O/h64_ab.txt
  FACEOFF_HIP_LIB=$PWD/$V/lib_$v.so python tools/bench_bf16.py "conv2_1 fwd" 2>&1 | grep -v "amdgpu.ids\|sum" >> $O/h64_ab.txt
done; done
cat $O/h64_ab.txt
