#!/bin/sh
# Writes tests/golden/benchmarker_rows_v1.csv and benchmarker_rows_v1_absdouble.csv: the rows the REFERENCE's own
# Benchmarker + CSV_Logger write for tests/golden/benchmarker_sequence.inc.  Runs in the build container only
# (/root/reference does not travel); the reference headers are compiled where they lie and as they stand - they are
# the only reference files that compile here (everything on the hot path needs fftw_cpp.hh).
#   plain g++:            `abs(elapsed - last)` (Benchmarker.hpp:66,104,127) finds only `int abs(int)`: differences
#                         truncated to whole milliseconds
#   g++ -include math.h:  the same unmodified headers with the double overload of abs in scope (what a toolchain whose
#                         <chrono>/<iostream> declare it, e.g. MSVC, compiles)
# The reference's stdout chatter is discarded; the CSV is the fixture.
set -e
here=$(cd "$(dirname "$0")" && pwd)
ref=${SOTS_REFERENCE:-/root/reference}
tmp=$(mktemp -d)
g++ -std=c++17 -O1 -I"$ref" -I"$here" -o "$tmp/plain" "$here/benchmarker_driver.cpp"
g++ -std=c++17 -O1 -include math.h -I"$ref" -I"$here" -o "$tmp/absdouble" "$here/benchmarker_driver.cpp"
"$tmp/plain" "$here/benchmarker_rows_v1.csv" > /dev/null
"$tmp/absdouble" "$here/benchmarker_rows_v1_absdouble.csv" > /dev/null
rm -rf "$tmp"
echo "wrote $here/benchmarker_rows_v1.csv $here/benchmarker_rows_v1_absdouble.csv"
