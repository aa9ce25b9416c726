#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 300 python tools/decoder_phases.py 2>/dev/null | head -30
