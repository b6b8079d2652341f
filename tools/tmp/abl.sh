for v in base sb1 sb2; do
  if [ $v = base ]; then unset RESNET_MI_LIB; else export RESNET_MI_LIB=variants/libresnet_mi_$v.so; fi
  echo "== $v"; python tools/bench_ops.py --bf16 --only fwd,dgrad 2>&1 | grep -v "^per-step\|^layer"
done
