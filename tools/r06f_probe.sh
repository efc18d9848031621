O=gpurun_out/r06f; mkdir -p $O
V=$PWD/faceoff_amd/csrc/variants
python -m pytest tests/test_lpips_gpu.py tests/test_bf16_ops_gpu.py tests/test_c3_gpu.py tests/test_bf16_engine_gpu.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
FACEOFF_HIP_LIB=$V/lib_stamp.so python tools/stamp_h64.py 2>&1 | grep -v amdgpu.ids | tee $O/stamp_h64.txt
for round in 1 2; do
for v in old noepi new; do
  echo "== $v (round $round)" >> $O/h64_ab.txt
  L=$V/lib_$v.so; [ $v = new ] && L=$PWD/faceoff_amd/libfaceoff_hip.so
  FACEOFF_HIP_LIB=$L python tools/bench_bf16.py conv1_2 2>&1 | grep -v "amdgpu.ids\|sum" >> $O/h64_ab.txt
  FACEOFF_HIP_LIB=$L python tools/bench_bf16.py "conv2_1 fwd" 2>&1 | grep -v "amdgpu.ids\|sum" >> $O/h64_ab.txt
done; done
cat $O/h64_ab.txt
for v in old new; do
  L=$V/lib_$v.so; [ $v = new ] && L=$PWD/faceoff_amd/libfaceoff_hip.so
  FACEOFF_HIP_LIB=$L python tools/soak_c3.py 100 > $O/soak_$v.txt 2>&1; grep "step 50\|step 100" $O/soak_$v.txt
done
