for n in 16384 8192 4096 2048; do
  case $n in 16384) NUMS=1,32,64,96,128,192,256,384,512,768;; 8192) NUMS=1,64,128,192,256,384,512,768,1024,1536;; 4096) NUMS=1,128,256,384,512,768,1024,1536,2048,3072;; 2048) NUMS=1,256,512,768,1024,1536,2048,3072,4096,6144;; esac
  echo "== n=$n latency kernels"; N=$n NUMS=$NUMS MI355NTT_LATENCY_PATH_MAX=1000000 python tools/crossover.py 2>/dev/null
  echo "== n=$n single-pass kernels"; N=$n NUMS=$NUMS MI355NTT_LATENCY_PATH_MAX=0 python tools/crossover.py 2>/dev/null
done
