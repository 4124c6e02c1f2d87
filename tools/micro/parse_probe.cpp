#include "fastx.hpp"
#include <chrono>
#include <cstdio>
#include <vector>
using namespace bronko;
int main(int argc, char** argv) {
    unsigned t = argc > 2 ? atoi(argv[2]) : 8;
    auto t0 = std::chrono::steady_clock::now();
    GzLineReader in(argv[1], t);
    std::string buf; std::vector<uint64_t> off{0};
    uint64_t n = 0, bases = 0;
    for (uint64_t ln = 0;; ln++) {
        if ((ln & 3) != 1) { if (!in.skip_next()) break; continue; }
        if (!in.append_next(buf)) break;
        off.push_back(buf.size());
        if (++n % 65536 == 0) { bases += buf.size(); buf.clear(); off.clear(); off.push_back(0); }
    }
    bases += buf.size();
    auto t1 = std::chrono::steady_clock::now();
    printf("%llu reads %llu bases %.3f s\n", (unsigned long long)n, (unsigned long long)bases, std::chrono::duration<double>(t1 - t0).count());
}
