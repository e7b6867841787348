# GPU tests + one bench line per workload (no CPU leg); results in gpurun_out/t.log and gpurun_out/b_new.log
rm -f gpurun_out/b_new.log
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
for w in cornell suzanne spheres; do timeout -k 10 200 python bench.py --workload $w --no-cpu --steps 2 >> gpurun_out/b_new.log 2>&1 || exit 1; done
timeout -k 10 300 python bench.py --workload final --spp 400 --no-cpu --steps 1 >> gpurun_out/b_new.log 2>&1
