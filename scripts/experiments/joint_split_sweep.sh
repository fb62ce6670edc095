cd $GRAFT_REPO_ROOT
for s in 0 2 3 4; do
  if [ $s = 0 ]; then unset TCSFM_JOINT_SPLIT; else export TCSFM_JOINT_SPLIT=$s; fi
  python scripts/dense_ref_timing.py ref_fi 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    print('split $s %-8s S=%d %-30s us/window %7.1f' % (d['HxW'], d['S'], d['launch'][:30], d['us_per_window']))"
done
