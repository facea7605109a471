// Exercises include/dtl_fec.hpp exactly the way the reference's C++ code uses its FEC objects
// (make_ldpc_encoders/decoders, fec_enc::encode, fec_dec::decode): prints one line per codeword:
// "<nit> <k decoded bits as 0/1 string>" for soft bits read from stdin (n floats per codeword).
#include "dtl_fec.hpp"

#include <cstdio>
#include <iostream>

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s <alist> [encode]\n", argv[0]); return 2; }
    try {
        auto encs = gr::dtl::make_ldpc_encoders({ argv[1] });
        auto decs = gr::dtl::make_ldpc_decoders({ argv[1] });
        if (encs[0] != nullptr || decs[0] != nullptr) return 3; // index 0 == "no FEC"
        const int n = decs[1]->get_n(), k = decs[1]->get_k();
        if (encs[1]->get_n() != n || encs[1]->get_k() != k) return 4;
        if (argc > 2) { // encode mode: k bits per line on stdin
            std::string line;
            std::vector<unsigned char> in(k), out(n);
            while (std::getline(std::cin, line) && int(line.size()) >= k) {
                for (int i = 0; i < k; i++) in[i] = line[i] == '1';
                encs[1]->encode(in.data(), k, out.data());
                for (int i = 0; i < n; i++) std::putchar(out[i] ? '1' : '0');
                std::putchar('\n');
            }
            return 0;
        }
        std::vector<float> llr(n);
        std::vector<unsigned char> bits(k);
        while (true) {
            for (int i = 0; i < n; i++) if (!(std::cin >> llr[i])) return 0;
            int nit = -1;
            if (decs[1]->decode(llr.data(), &nit, bits.data()) != k) return 5;
            std::printf("%d ", nit);
            for (int i = 0; i < k; i++) std::putchar(bits[i] ? '1' : '0');
            std::putchar('\n');
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
