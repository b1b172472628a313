# After tools/final_validation.sh (gpurun has merged gpurun_out/ back): the summaries that are kept, into profiles/ (tracked).
set -eu
cd "$(dirname "$0")/.."
R=${ROUND:-r06}
cp gpurun_out/prof/${R}_kernel_stats.csv gpurun_out/prof/${R}_pmc.csv profiles/
python tools/make_traffic.py ${R}
for c in sarsa c2 double_q c5 eps01; do
  cp gpurun_out/prof/${R}_${c}_kernel_stats.csv gpurun_out/prof/${R}_${c}_pmc.csv profiles/
  python tools/make_traffic.py ${R}_${c} profiles/pmc_traffic_${c}.json > /dev/null
done
mkdir -p profiles/${R}_bench
cp gpurun_out/bench_${R}_*.json profiles/${R}_bench/
cp gpurun_out/gputest_${R}.log gpurun_out/sweep_cut_short.txt gpurun_out/exp_upload_${R}.txt profiles/${R}_bench/ 2>/dev/null || true
python tools/kernel_resources.py --csv profiles/${R}_resources.csv > /dev/null
ls profiles | grep ${R}
