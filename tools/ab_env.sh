#!/bin/bash
# A/B of environment-variable variants inside ONE gpurun call: tools/ab_env.sh "<bench args>" VAR=a VAR=b ...   (two rounds each)
ARGS=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in "$@"; do
    echo -n "$v: "
    env $v python bench.py $ARGS --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=d['stages']
print(d['value'], d['ms_per_step'], {k:{kk:vv['ms'] for kk,vv in v.get('breakdown',{}).items() if kk in ('gemm','mlp_fused','layernorm','attn_window','attn_global')} for k,v in st.items()})"
  done
done
