#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for e in 0 1; do echo "=== GJ_DEC_NO_BALANCE=$e"; GJ_DEC_NO_BALANCE=$e timeout 300 python tools/decoder_phases.py 2>/dev/null | sed -n 1,31p | grep -v "^  at "; done
