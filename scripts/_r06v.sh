cd /root/repo
O=gpurun_out/r06z; mkdir -p $O
ACGPU_LIB=/root/repo/aho-corasick_amd/lib/exp/libacgpu_pfx_prof.so timeout 300 python scripts/pfx_prof.py > $O/pfx_prof.jsonl 2> $O/err.txt
cat $O/pfx_prof.jsonl; tail -3 $O/err.txt
