#!/bin/bash
cd tests
IHP_DEBUG_SYNC=1 timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -20
import sys
sys.path.insert(0,'..')
import indelope_amd, golden_util
api=indelope_amd.api(); api.init(0)
golden_util.check_regions(api,'long')
print('ok')
PY
