# backward-data shapes of the training step that run far above their forward twins: which table config / split-K is fastest
for sh in "10,1024,512,3,1" "10,512,1024,3,1" "20,512,256,3,1" "160,64,32,3,1"; do
  python scripts/dev/bench_conv.py --shapes custom --custom $sh --cfgs=-1,0,1,2,3,5,6,10,11,12,13,17,27 --nores --reps 10 2>/dev/null | tail -1
  for sk in 2 3 4; do
    python scripts/dev/bench_conv.py --shapes custom --custom $sh --cfgs=-1,0,2,3,13 --nores --reps 10 --splitk $sk 2>/dev/null | tail -1
  done
done
