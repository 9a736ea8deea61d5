cd /root/repo
O=gpurun_out/r06w; mkdir -p $O
E=/root/repo/aho-corasick_amd/lib/exp
for i in 1 2; do
  echo "ring256" >> $O/summary.txt; timeout 300 python scripts/bench_nat.py 20 2>> $O/nat.err | cut -c1-200 >> $O/summary.txt
  echo "ring128" >> $O/summary.txt; ACGPU_LIB=$E/libacgpu_pfx_ring128.so timeout 300 python scripts/bench_nat.py 20 2>> $O/nat.err | cut -c1-200 >> $O/summary.txt
done
ACGPU_LIB=$E/libacgpu_pfx_ring128_prof.so timeout 300 python scripts/pfx_prof.py >> $O/summary.txt 2>> $O/nat.err
ACGPU_LIB=$E/libacgpu_pfx_ring128.so scripts/pmc_traffic.sh $O/nat_sherlock_ring128_pmc.json "k_pfx_count<true" 1 "sherlock 1 GiB / words-5000, rings of 128" -- python /root/repo/scripts/bench_nat.py 4 sherlock >> $O/summary.txt 2>&1
cat $O/summary.txt
