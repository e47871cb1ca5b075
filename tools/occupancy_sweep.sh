for pad in 0 1400 2700 5300 8000; do
  for wl in c2 peaks; do
    echo "pad $pad $wl: $(GKM_LDS_PAD=$pad python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-also 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["parity"]["ok"])')"
  done
done
