for ab in 0 3 7 11 19 31 15 27; do echo "ABLATE $ab"; FF_LOOKUP_ABLATE=$ab ONLY=lookup python tools/bench_lookup.py 2>&1 | grep warm; done
for w in 4 8 16; do echo "WPC $w ABL 3"; FF_LOOKUP_WAVES_PER_CU=$w FF_LOOKUP_ABLATE=3 ONLY=lookup python tools/bench_lookup.py 2>&1 | grep warm; done
