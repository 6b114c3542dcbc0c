#!/bin/bash
# One 16-hex-digit build id: SHA-256 over the CODE of the given sources (comments and blank lines removed: the C++ preprocessor run on
# already-preprocessed input keeps every token, directive and macro definition and drops only the comments — an edit to a comment no
# longer orphans the PMC summaries under profiles/), the variant's flag string and the compiler's version.
#   build_id.sh "<flags and compiler version>" <source> ...
set -o pipefail
tag="$1"; shift
{
  for f in "$@"; do
    echo "== $(basename "$f")"
    ${CXX:-g++} -fpreprocessed -dD -E -P -x c++ "$f" 2>/dev/null || { echo "build_id.sh: cannot strip $f" >&2; exit 1; }
  done
  echo "$tag"
} | sha256sum | cut -c1-16
