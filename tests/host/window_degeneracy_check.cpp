// Host logic check (no GPU): the facade's evalDegenracy(local_param_ids, WindowNormalEquations, frame_cnt, state) -- Estimator::evalDegenracy
// (estimator.cpp:1598-1680) over the C-ABI's host helper -- on a J^T J read from a file. Compiled and run by
// tests/test_abi.py::test_facade_window_degeneracy_policy, which holds the output against the CPU restatement (and, through
// tests/test_oracle_ref_pin.py, against the reference's own lines).
// argv: dir  D  opt_window_size  num_of_laser  estimate_extrinsic  frame_cnt  n_cumu_feature  lambda_thre_calib
#include "mloam_facade.hpp"
#include <cstdio>
#include <cstdlib>
#include <fstream>

template <typename T> static std::vector<T> read_file(const std::string &p)
{
    std::ifstream f(p, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot read %s\n", p.c_str()); std::exit(2); }
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<T> v(size_t(n) / sizeof(T));
    f.read(reinterpret_cast<char *>(v.data()), n);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 9) return 2;
    const std::string d = std::string(argv[1]) + "/";
    using namespace mloam_hip;
    Params &P = params();
    WindowNormalEquations ne;
    ne.D = std::atoi(argv[2]);
    P.OPT_WINDOW_SIZE = std::atoi(argv[3]); P.NUM_OF_LASER = std::atoi(argv[4]); P.ESTIMATE_EXTRINSIC = std::atoi(argv[5]);
    const int frame_cnt = std::atoi(argv[6]);
    P.N_CUMU_FEATURE = std::atoi(argv[7]); P.LAMBDA_THRE_CALIB = std::atof(argv[8]);
    ne.JtJ = read_file<double>(d + "JtJ.f64");
    ne.n_residuals = 1;
    WindowDegeneracyState st;
    st.eig_thre = read_file<double>(d + "eig_thre.f64");
    const size_t nb = size_t(ne.D) / 6;
    std::vector<PoseLocalParameterization> store(nb);
    std::vector<PoseLocalParameterization *> ids;
    for (auto &p : store) { p.setParameter(); ids.push_back(&p); }
    evalDegenracy(ids, ne, frame_cnt, st);
    std::vector<double> out;
    for (size_t i = 0; i < nb; ++i) {
        out.push_back(store[i].is_degenerate_ ? 1.0 : 0.0);
        out.push_back(st.eig_thre[i]);
        out.insert(out.end(), store[i].V_update_.begin(), store[i].V_update_.end());
    }
    for (double x : st.d_factor_calib) out.push_back(x);
    std::ofstream f(d + "out.f64", std::ios::binary);
    f.write(reinterpret_cast<const char *>(out.data()), std::streamsize(out.size() * sizeof(double)));
    return 0;
}
