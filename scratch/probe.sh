#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
cp syncvsr_amd/libsyncvsr_hip.so /tmp/lib_backup.so
for f in syncvsr_amd/csrc/*.hip; do
  o=/tmp/$(basename $f .hip).o
  extra="-DSVSR_PROBE"
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=fast $extra -c $f -o $o &
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o syncvsr_amd/libsyncvsr_hip.so /tmp/*.o
python scratch/probe_ep.py
