#!/bin/bash
# round 4, GPU call 17: rocprofv3 passes behind the WARP roofline again (configs[4] size now run for 12 epochs like the bench: T = 1.8)
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/gpu_profile_warp.sh 2>&1 | tail -30 | cut -c1-250
