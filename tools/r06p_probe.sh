O=gpurun_out/r06p; mkdir -p $O
python tools/probes/convT_cells_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/cells_probe.txt
python -m pytest tests/test_bench_gpu.py -m gpu -q -x > $O/t.log 2>&1; tail -4 $O/t.log
