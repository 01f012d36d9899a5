cd /root/repo; export PYTHONPATH=$PWD PRECISE_MODES=0 SPIDER_GEMM_TRACE=1
timeout -k 10 300 python3 scripts/exp/precise_cost.py zeroscope 2> gpurun_out/r05_trace3d.err | grep "ms per"
grep spider_gemm_dispatch gpurun_out/r05_trace3d.err | sort | uniq -c | sort -rn | head -70 > gpurun_out/r05_unet3d_dispatch2.txt
rm gpurun_out/r05_trace3d.err
