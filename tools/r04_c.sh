out=$PWD/gpurun_out/r04c; mkdir -p $out; repo=$PWD
bash tools/prof_generate.sh r04c/gen 20 16 > $out/gen_prof.log 2>&1; tail -45 $out/gen_prof.log | cut -c1-170
cd /tmp && export TMPDIR=/tmp
SQ="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout 300 rocprofv3 --pmc $SQ --kernel-trace -d $out -o gemm_sq --output-format csv -- python $repo/tools/probe_gemm_pmc.py > $out/gemm_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out -o gemm_fetch --output-format csv -- python $repo/tools/probe_gemm_pmc.py > $out/gemm_fetch.log 2>&1
cd $repo
f=$(find $out -name "gemm_sq_counter_collection.csv" | head -1); python tools/pmc_kernel_means.py $f k_gemm
f=$(find $out -name "gemm_fetch_counter_collection.csv" | head -1); python tools/pmc_kernel_means.py $f k_gemm
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete; find $out -name "*counter_collection.csv" -size +20M -delete
