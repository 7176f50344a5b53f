#!/bin/bash
# kernel time of the dense StereoBM drop-in (svo_stereo_bm) at 1241x376, 48 disparities, 21x21
set -euo pipefail
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_dense"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
cat > "$OUT/run.py" <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, numpy as np
import stereo_vo_amd as S
ctx = S.Context(1241, 376)
sp = S.api.synth_default_params() if hasattr(S.api, "synth_default_params") else None
rng = np.random.default_rng(1)
tex = np.kron(rng.integers(0, 255, (130, 440)), np.ones((3, 3))).astype(np.uint8)
L = np.ascontiguousarray(tex[:376, 20:20 + 1241]); R = np.ascontiguousarray(tex[:376, 27:27 + 1241])
for _ in range(5): d = ctx.stereo_bm(L, R)
print("valid fraction", float((d > 0).mean()))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o p -- python3 "$OUT/run.py" > "$OUT/log.txt" 2>&1
tail -2 "$OUT/log.txt" | head -1
grep -i "stereo" "$OUT/p_kernel_stats.csv" | cut -c1-40,150-260
