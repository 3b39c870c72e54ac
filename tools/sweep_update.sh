#!/bin/bash
# sweep UCE_UPDATE_VARIANT over the update kernels (one process per variant: the choice is read once)
for v in 20 26 27; do UCE_UPDATE_VARIANT=$v timeout 120 python tools/probe_update.py 16,32,50,64,100,128 2>&1 | tail -1; done
