#!/bin/bash
# round 4, GPU call S: fabric reads of the headline kernel (PMC pass over --no-also), per-start table with the small trie
# in LDS (parity + the dense definitions)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04s; mkdir -p $O
BENCH_ARGS=--no-also PASSES=tcc3 PMC_GIB=8 timeout 200 scripts/gpu_pmc.sh > $O/pmc_pf.log 2>&1; tail -2 $O/pmc_pf.log | cut -c1-300
d=$(ls -d gpurun_out/pmc_[0-9]* | tail -1); python scripts/pmc_to_json.py "$d" "k_pf_count<" $O/pf_pmc.json "per-dispatch averages of k_pf_count<false,false>, headline workload 8 GiB (bench.py --no-also: every launch is a headline launch); rocprofv3 --pmc TCC_EA0_RDREQ* pass (scripts/gpu_pmc.sh)" | tail -4
timeout 600 python -m pytest tests/test_gpu_find_dense.py tests/test_gpu_find.py tests/test_gpu_bench_defs.py tests/test_gpu_golden.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
timeout 300 python scripts/bench_defs.py 256 auto -match,-common,earlyshort 2>&1 | grep '"bench"' | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['family'], r['bench'], 'ov', r.get('ov_call_GBps'), 'lf', r.get('lf_find_iter_GBps'), 'cpu_lf', r.get('cpu_lf_GBps'))" | tee $O/defs.txt
