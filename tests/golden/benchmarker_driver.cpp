// Feeds tests/golden/benchmarker_sequence.inc to whichever Benchmarker.hpp the include path finds:
// the reference's (tests/golden/make_benchmarker_golden.sh, build container only) or this repository's
// host/Benchmarker.hpp (tests/test_host_cpu.py).  argv[1] = the CSV to write.
#include <cstdint>
#include <string>
#include <vector>

#include "Benchmarker.hpp"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    // the field list every backend of the reference passes (Evolutionary_Strategy_OpenCL.hpp:117)
    Benchmarker b(argv[1], {"Test_Name", "Total_Time", "Average_Time", "Max_Time", "Min_Time", "Max_Difference", "Average_Difference"});
#include "benchmarker_sequence.inc"
    return 0;
}
