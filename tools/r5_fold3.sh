cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
lib=$GRAFT_REPO_ROOT/representationlearning_amd/lib/ab/librssf_$v.so
echo "== $v"
for i in 1 2 3; do RSSF_LIB_OVERRIDE=$lib timeout 300 python tools/attn_c48_dbg.py 2>&1 | grep -E '^[01] (gx|gy|attn.v_proj.weight|weight_levels.weight)' | awk '{printf "%s:%s %s/%s ; ", $1, $2, $5, $7}'; echo; done
done
