#!/bin/bash
# round 6, session 3: guarded sin/cos of the normal-map sampling scheme -- whole GPU suite (bit-identity), then the legs it touches
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=$R/gpurun_out/r06; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests3.log 2>&1; echo "pytest rc $?" >> $O/gputests3.log
tail -4 $O/gputests3.log
for w in tabular_sample tabular_abc_sample tabular_aniso_sample; do
  python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 10 --warmup 5 > $O/bench3_$w.json 2>/dev/null
  python -c "import json; d=json.loads(open('$O/bench3_$w.json').read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['roofline']['frac'])"
done
