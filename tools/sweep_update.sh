#!/bin/bash
# sweep UCE_UPDATE_VARIANT over the update kernels (one process per variant: the choice is read once)
# 20 = 16-row tiles (default), 21 = 32-row tiles
for v in 20 21; do UCE_UPDATE_VARIANT=$v timeout 120 python tools/probe_update.py 16,32,50,64,100,128 2>&1 | tail -1; done
