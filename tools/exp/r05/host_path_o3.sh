#!/bin/bash
# round 5 (last session): one-pair calls of the djb:: surface -- the reference's real callers -- timed beside the REAL reference
# (oracle/_ref/scalar_latency = the same source on /root/reference/dj_brdf.h, -O2) on the GPU box's host cores:
# a GPU object answered by its host twin, the CPU context, and the reference, alternating, three passes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; O=gpurun_out/host_path_o3.txt
{
echo "# ns per call: ggx.eval(i, o, &params) | ggx.pdf(i, o) | beckmann.sample(u1, u2, o, &params) | tabular.evalp(i, o) | beckmann.evalp_is | tabular.evalp_is | eval through a brdf*, independent calls | the same, each call depending on the previous result; then M calls/s of 16 threads on one object"
for rep in 1 2 3; do
  for v in "gpu-object(host twin)" "cpu-context" reference; do
    case $v in gpu*) cmd="./examples/scalar_latency 16";; cpu*) cmd="env DJB_DEVICE=cpu ./examples/scalar_latency 16";; *) cmd="./oracle/_ref/scalar_latency 16";; esac
    DJB_QUIET=1 $cmd 2>/dev/null | awk -v v="$v" '/ns per call/ { printf "%s ", $(NF-3) } /threads on one/ { printf "| %s M calls/s", $7 } END { printf "   (%s)\n", v }'
  done
done
} > $O 2>&1
cat $O
