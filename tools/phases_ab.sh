#!/bin/bash
# phase stamps of k_huffman_decode_tok for several trace builds: tools/phases_ab.sh lib1 lib2 ... (names under gpujpeg_amd/lib)
cd "${GRAFT_REPO_ROOT:-.}"
for lib in "$@"; do
  echo "=== $lib"
  GJ_TRACE_LIB=gpujpeg_amd/lib/$lib timeout 300 python tools/decoder_phases.py 2>&1 | grep -v "^k_markers\|^  at " | head -32
done
