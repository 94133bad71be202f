python tools/bench_pw.py 2>&1 | grep cin | head -3
MSL_PW_STRIP_NARROW=1 python tools/bench_pw.py 2>&1 | grep cin | head -3
