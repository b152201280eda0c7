#!/bin/bash
# end-of-round session: all GPU tests, smoke(), the driver's bench command, profiles, the other two workloads
R=$(pwd); O=$R/gpurun_out/r02; mkdir -p $O
bash tools/gpu_r02_session.sh tests bench prof
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 300 python bench.py --workload tetra > $O/bench_tetra.json 2> $O/bench_tetra.err; echo "tetra rc=$?"; cut -c1-400 $O/bench_tetra.json
timeout 600 python bench.py --workload anib > $O/bench_anib.json 2> $O/bench_anib.err; echo "anib rc=$?"; cut -c1-300 $O/bench_anib.json
