#!/bin/bash
# Untraced A/B of the bare train loop under environment switches (GPU box, repo root):
#   tools/ab_env.sh "<label>:<VAR=value,VAR=value>" ...   -> one ms/step line per configuration, each run twice
for spec in "$@"; do
  label=${spec%%:*}; vars=${spec#*:}
  for rep in 1 2; do
    ( IFS=','; for kv in $vars; do [ -n "$kv" ] && export "$kv"; done
      printf '%-28s ' "$label"; timeout -k 10 200 python3 tools/train_loop.py ${AB_STEPS:-30} ${AB_BATCH:-32} ${AB_BASE:-hg2} | tail -1 ) || exit 1
  done
done
