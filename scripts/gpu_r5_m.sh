#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python scripts/probes/halo_probe.py 2>&1 | tail -14
