# early layers of the TRAINING forward / backward-data (unfused): which table config is fastest
for sh in "320,32,64,3,2" "160,32,64,3,1" "160,64,32,3,1" "160,64,32,1,1" "160,32,64,1,1" "160,64,128,3,2" "80,128,64,1,1" "80,64,128,1,1" "80,64,128,3,1" "80,128,64,3,1"; do
  python scripts/dev/bench_conv.py --shapes custom --custom $sh --cfgs=-1,0,1,2,3,5,6,10,11,12,13,17,27 --nores --reps 10 2>/dev/null | tail -1
done
