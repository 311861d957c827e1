TAG=r06s3 PYTEST_ARGS="tests/test_gpu_flat.py tests/test_gpu_scale.py tests/test_gpu_random.py tests/test_gpu_full_size.py" PYTEST_K="flat or Flat" tools/gpu_round.sh tests
bash tools/flat_ab.sh main > gpurun_out/r06s3_flat.txt 2>&1
cut -c1-330 gpurun_out/r06s3_flat.txt
EXP_ARGS="--layouts 2 --rounds 2 --steps 10" tools/ab_variants.sh main v2 v3 v2r4 v3r4 2>&1 | tee gpurun_out/r06s3_scan_stream.txt
