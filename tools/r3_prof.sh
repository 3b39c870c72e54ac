bash tools/prof_round.sh r03h sd14_erase50 sd14_erase2p3 sd14_erase100 sd14_erase1000p500 sdxl_debias36x2 xattn sattn > gpurun_out/r03h_prof.log 2>&1
tail -3 gpurun_out/r03h_prof.log
