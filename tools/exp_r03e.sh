for p in 1 2 3 4 5; do for v in c5 w_f; do echo "== $v (process $p)"; KB_PAIR=1 KB_B2B=4 ./tools/kbench_$v 1024 40 20 30; done; done
