mkdir -p gpurun_out/r6a
for T in 1 100 200; do AB_PASSES=2 python tools/option_ab.py nt_out 0 $T zzz > gpurun_out/r6a/nt_out_$T.txt 2>&1; grep -v "pass\|kernels:" gpurun_out/r6a/nt_out_$T.txt; done
