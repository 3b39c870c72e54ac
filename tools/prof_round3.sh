#!/bin/bash
# round-3 profile pass: rocprofv3 stats + PMC of every bench workload and attention leg, chain timelines, generation profile
bash tools/prof_round.sh ${1:-r03} sd14_erase50 sd14_erase2p3 sd14_erase100 sd14_erase1000p500 sdxl_debias36x2 xattn sattn > gpurun_out/${1:-r03}_prof.log 2>&1
tail -5 gpurun_out/${1:-r03}_prof.log
for wl in sd14_erase50 sd14_erase100; do UCE_CHAIN_DEBUG=1 timeout 300 python tools/dbg_chain.py $wl > gpurun_out/${1:-r03}/chain_stamps_$wl.txt 2>&1; done
UCE_CHAIN_DEBUG=1 timeout 300 python tools/dbg_potrf.py > gpurun_out/${1:-r03}/potrf_walker_stamps.txt 2>&1
