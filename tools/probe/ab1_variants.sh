# same-box comparison of attention-backward builds (tools/probe/build/lib_<name>.so), encoder shape
for v in "" $@; do
  if [ -z "$v" ]; then echo "== default"; python tools/probe/attn_bwd_ab.py 2>/dev/null | grep -E "^one|^fwd";
  else echo "== $v"; NS_LIB_PATH=tools/probe/build/lib_$v.so python tools/probe/attn_bwd_ab.py 2>/dev/null | grep -E "^one|^fwd"; fi
done
