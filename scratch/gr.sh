#!/bin/bash
# gpurun wrapper: retries only the "no box free" answer (exit 3, nothing ran, nothing charged)
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
