O=gpurun_out/r06g; mkdir -p $O
V=$PWD/faceoff_amd/csrc/variants
for round in 1 2; do
for v in new nodma nomfma nostore nomem; do
  echo "== $v (round $round)" >> $O/h64_ablate.txt
  L=$V/lib_$v.so; [ $v = new ] && L=$PWD/faceoff_amd/libfaceoff_hip.so
  FACEOFF_HIP_LIB=$L python tools/bench_bf16.py "conv1_2 fwd" 2>&1 | grep -v "amdgpu.ids\|sum" >> $O/h64_ablate.txt
done; done
cat $O/h64_ablate.txt
