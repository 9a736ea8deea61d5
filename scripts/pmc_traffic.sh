#!/bin/bash
# Fabric read traffic of ONE kernel of any command: a single rocprofv3 --pmc pass (TCC_EA0_RDREQ by request size; --kernel-trace
# only, as the pool requires) -> profiles-ready JSON {hbm_read_bytes per dispatch, _kernel, _gib, _note}.
# usage: pmc_traffic.sh <out.json> <kernel substring> <algorithmic GiB per launch> <note> -- <command...>
set -u
cd "$(dirname "$0")/.."
OUTJ=$1; KERN=$2; GIB=$3; NOTE=$4; shift 5
export TMPDIR=/tmp
ROOT=$PWD
D=gpurun_out/pmc_traffic_$$
mkdir -p "$D"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum \
    --output-format csv -d "$ROOT/$D/tc3" -o pmc -- "$@" > "$ROOT/$D/cmd.out" 2> "$ROOT/$D/cmd.err")
echo "pmc pass exit $? ($KERN)"
python scripts/pmc_to_json.py "$D" "$KERN" "$OUTJ" "$NOTE" "$GIB" > /dev/null
rm -rf "$D"
python - "$OUTJ" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("_kernel", "_gib", "hbm_read_bytes", "_dispatches")})
PY
