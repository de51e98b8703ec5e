for p in 1 2 3; do for v in c4 v_lf; do echo "== $v (process $p)"; KB_PAIR=1 KB_B2B=4 ./tools/kbench_$v 1024 40 20 30; done; done
