export TAG=r06last
tools/gpu_round.sh env pmc_fetch
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
tools/gpu_round.sh tests smoke bench prof per_rank latency flat
