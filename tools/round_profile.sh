#!/bin/bash
# One GPU-box pass that produces everything profiles/ holds for a round:  tools/round_profile.sh <tag>
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd $ROOT
python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15 > $O/${TAG}_pytest_gpu.txt
python bench.py > $O/${TAG}_bench.json 2> $O/bench.err
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python $ROOT/bench.py --steps 3 --no-cpu --no-per-rank --no-opq --no-traffic > $O/bench_prof.out 2>&1)
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_bench_steps3.csv
grep -h "^{\"metric\"" $O/bench_prof.out > $O/${TAG}_bench_under_rocprof.json   # the bench line of the PROFILED run (its HIP-event average must match the CSV)
tools/pmc_collect.sh $O/${TAG}_pmc_counters_raw.json -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-per-rank --no-opq --no-traffic --adc-batches 1 > $O/pmc.out 2>&1
tools/pmc_collect.sh $O/${TAG}_pmc_kmeans_raw.json -- python $ROOT/tools/kmeans_one.py 48 > $O/pmc_km.out 2>&1
python tools/pmc_summary.py $O/${TAG}_pmc_counters_raw.json $O/pmc_summary.json "MI355X, round ${TAG:2:1}, profile $TAG." $O/${TAG}_pmc_kmeans_raw.json >> $O/pmc.out 2>&1
python tools/config_bench.py 2>&1 | grep constrained > $O/${TAG}_config_bench.txt
python tools/adc_quick_bench.py 48 96 64 32 16 2>&1 | grep QPS >> $O/${TAG}_config_bench.txt
python tools/kmeans_bench.py 48 96 64 24 8 2>&1 | grep kmeans_stats >> $O/${TAG}_config_bench.txt
python tools/ivf_bench.py 96 2>&1 | grep "nprobe=" >> $O/${TAG}_config_bench.txt
# per-kernel tables of the searches (the kernels AROUND the screens) and of the index build
for np in 8 32 128; do echo "== IVF M=96 nlist=5000 nprobe $np (us per 1200-query search)"; tools/ivf_one_prof.sh $np; done > $O/${TAG}_search_kernels.txt 2>&1
echo "== flat M=48 (us per launch; 8 launches = 8 batches of 1200 queries)" >> $O/${TAG}_search_kernels.txt; tools/adc_prof.sh 48 2>&1 | grep -v "at::native" >> $O/${TAG}_search_kernels.txt
echo "== index build 2^20 x 768, M=48" >> $O/${TAG}_search_kernels.txt; tools/nearest_prof.sh 1048576 48 >> $O/${TAG}_search_kernels.txt 2>&1
# counters of the side kernels and of the k-means statistics
tools/pmc_collect.sh $O/${TAG}_pmc_ivf_nprobe8_raw.json -- python $ROOT/tools/ivf_one.py 8 > $O/pmc_ivf.out 2>&1
python tools/first_contact.py 2 > $O/${TAG}_first_contact.txt 2>&1
tail -3 $O/${TAG}_pytest_gpu.txt; cat $O/${TAG}_bench.json | cut -c1-600
