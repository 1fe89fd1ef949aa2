for x4 in 1 0; do for tile in 1024 2048 4096 8192; do for bpc in 4 8 16; do
  r=$(AUKIT_FAST_TILE=$tile AUKIT_FAST_BLOCKS_PER_CU=$bpc python bench.py --steps 10 --warmup 2 --cpu-streams 0 --store-x4 $x4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e3,1), round(d['roofline']['achieved']), round(d['roofline']['kernel_ms'],3))")
  echo "x4=$x4 tile=$tile bpc=$bpc -> $r"
done; done; done
