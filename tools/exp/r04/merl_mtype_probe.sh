#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
PYTHONPATH=. timeout 900 python tools/merl_mtype_probe.py > gpurun_out/merl_mtype_probe.txt 2>&1; cat gpurun_out/merl_mtype_probe.txt
