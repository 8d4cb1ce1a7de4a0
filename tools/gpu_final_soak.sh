#!/bin/bash
# closing soak of the library as committed: fresh seeds (2000 .. 6399) of the sequence fuzzer, four threads, all profiles;
# then the same with every fusable batch on the generated DIRECT kernels and table images (tiny inputs, 4 partitions)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 330 python tools/stress_fuzz.py --iters 1100 --threads 4 --seeds 4400 --profiles --budget-s 190 --tag final_soak > gpurun_out/final_soak.json 2> gpurun_out/final_soak.err
echo "soak rc $?"; tail -c 1200 gpurun_out/final_soak.json
ARES_LEAN_MIN_GROUPS=0 ARES_MIN_PART_BITS=2 timeout 200 python tools/stress_fuzz.py --iters 400 --threads 4 --seeds 1600 --profiles --kernels --budget-s 70 --tag final_image_soak > gpurun_out/final_image_soak.json 2> gpurun_out/final_image_soak.err
echo "image soak rc $?"; tail -c 1500 gpurun_out/final_image_soak.json
# round 6: every Sort + Reduce of the "sort" profiles through the wide layout with two partition levels, blocks checked clean
ARES_MEM_VERIFY_CLEAN=1 ARES_SR_SCAN_FED=0 ARES_SRV_PART_BITS=11 timeout 200 python tools/stress_fuzz.py --iters 400 --threads 4 --seeds 1600 --profiles --kernels --budget-s 70 --tag final_wide_soak > gpurun_out/final_wide_soak.json 2> gpurun_out/final_wide_soak.err
echo "wide soak rc $?"; tail -c 1500 gpurun_out/final_wide_soak.json
