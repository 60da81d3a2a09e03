#!/bin/bash
# GPU box, round 5: WTA launch with the frame as the fastest block index (wtaz) on the other configurations that run wta_kernel; default placement search
R=$GRAFT_REPO_ROOT; cd $R
CONFIGS="c1 c2b8 c2" PARITY_VARS="" bash profiles/tools/r05_ab.sh r05_wtaz2 "base wtaz" 4
