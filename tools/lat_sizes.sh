for n in 2048 4096 8192 16384 32768; do for f in 1000000 0; do echo "n=$n MI355NTT_LATENCY_PATH_MAX=$f"; MI355NTT_LATENCY_PATH_MAX=$f ntt-cuda_amd/build/lat_bench 200 5 $n | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   ok', d['round_trip_ok'], d['bfv_round_trip_ok'], 'stream', d['batch1_stream_us'], 'bfv', d['bfv_4plus1_primes_stream_us'])"; done; done
