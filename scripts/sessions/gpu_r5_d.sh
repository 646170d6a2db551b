#!/bin/bash
# round 5, session D: kernel traces of the token stream (what does each kernel of stage 1 / the depth scan / the tape front take with and without it),
# the new GPU tests of the token-fed tape
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "token or stage2" --timeout 900 -p no:cacheprovider > $O/r5d_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r5d_pytest.log
for w in amazon_ndjson; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/prof_r5d_tok_$w -o t -- python $GRAFT_REPO_ROOT/scripts/tokens_once.py $w 1073741824 > $O/r5d_tok_$w.log 2>&1); echo "trace $w rc=$?"; tail -1 $O/r5d_tok_$w.log
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/prof_r5d_tok_twitter -o t -- python $GRAFT_REPO_ROOT/scripts/tokens_once.py twitter_like 268435456 > $O/r5d_tok_twitter.log 2>&1); echo "trace twitter rc=$?"; tail -3 $O/r5d_tok_twitter.log
python3 scripts/rocpd_summary.py $O/prof_r5d_tok_amazon_ndjson/*/t_results.db $O/prof_r5d_tok_amazon_ndjson/t_results.db 2>/dev/null | head -40 | cut -c1-150
python3 scripts/rocpd_summary.py $O/prof_r5d_tok_twitter/*/t_results.db $O/prof_r5d_tok_twitter/t_results.db 2>/dev/null | head -50 | cut -c1-150
