#!/bin/bash
# Runs the given commands (one per argument) in order on the GPU box.  A plain failure (exit 1: an assertion) lets the
# chain go on; a time limit, abort, kill or segmentation fault ends it there (nothing more is started on the GPU).
rc_all=0
for cmd in "$@"; do
  echo "=== $cmd"
  bash -o pipefail -c "$cmd"
  rc=$?
  if [ $rc -ne 0 ]; then
    rc_all=$rc
    echo "=== exit status $rc"
    case $rc in 124|134|137|139|143) echo "=== hard failure: stopping the chain"; exit $rc;; esac
  fi
done
exit $rc_all
