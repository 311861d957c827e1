cd "${GRAFT_REPO_ROOT:-/root/repo}"
export RSX_LIB=$PWD/retrieval-scaling_amd/csrc/librsx_measure.so
for v in 0 1 17 33 65 49 113; do
  RSX_ROT_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 3 --cpu-queries 0 --no-recall --no-configs --no-faiss > gpurun_out/var_$v.json 2> gpurun_out/var_$v.log
  echo "variant $v: $(python tools/show_bench.py gpurun_out/var_$v.json | head -1 | cut -c100-260)"
done
