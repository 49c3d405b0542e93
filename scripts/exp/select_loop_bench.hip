// Host-only timing of the good-feature selection loops (m-loam_amd/csrc/select.hip: select_rnd, select_greedy) on synthetic rows shaped like config 5's
// (11.7 k surf rows 97 % matched, 11.2 k corner rows 40 % matched, ratio 0.2). The translation unit is select.hip itself, so the loops timed are the
// product's; nothing here touches a GPU (the HIP symbols the file's launch helpers reference are left unresolved at link time and never called).
// Build + run: scripts/exp/select_loop_bench.sh. The checksum is over the selected indices: it must not move when the loops are changed.
// With --check nothing is timed: the random loop is compared with the container the reference uses (a std::vector it draws positions from and erases), and
// the greedy loop's determinant-lemma scoring with the literal Cholesky-logdet scoring (MLH_SELECT_EXACT), over sizes, matched fractions, ratios and seeds,
// rows repeated verbatim included (exact ties); tests/test_select_loops_host.py runs this on the CPU.
#include "../../m-loam_amd/csrc/select.hip"
#include <cstdio>
using namespace mlh;
static void make_rows(size_t m, double frac, unsigned seed, bool with_repeats, std::vector<uint8_t> &valid, std::vector<double> &J)
{
    valid.resize(m); J.resize(6 * m);
    std::mt19937 g(seed); std::uniform_real_distribution<double> u(-1, 1);
    for (size_t i = 0; i < m; ++i) {
        valid[i] = (u(g) * 0.5 + 0.5) < frac;
        double n[3] = {u(g), u(g), u(g)}; const double nn = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]); for (double &x : n) x /= nn;
        const double p[3] = {30 * u(g), 30 * u(g), 3 * u(g)}, w = 1.0 + u(g) * 0.5;
        J[6*i+0]=w*n[0]; J[6*i+1]=w*n[1]; J[6*i+2]=w*n[2]; J[6*i+3]=w*(p[1]*n[2]-p[2]*n[1]); J[6*i+4]=w*(p[2]*n[0]-p[0]*n[2]); J[6*i+5]=w*(p[0]*n[1]-p[1]*n[0]);
        if (with_repeats && i >= 8 && (i % 3) == 0) for (int c = 0; c < 6; ++c) J[6*i+c] = J[6*(i-7)+c];     // a third of the rows repeat an earlier one: exact ties in the scores
    }
}
// the random loop as the reference has it (lidar_mapper.h:300-345): positions drawn in, and elements erased from, a std::vector
static void rnd_literal(const Rows &R, size_t n_use, std::mt19937 &rng, std::vector<size_t> &sel, double H[36])
{
    std::vector<size_t> all(R.size());
    std::iota(all.begin(), all.end(), size_t(0));
    while (sel.size() < n_use && !all.empty()) {
        const size_t j = std::uniform_int_distribution<size_t>(0, all.size() - 1)(rng);
        const size_t q = all[j];
        if (R.matched(q)) { rank1_update(H, R.jaco(q)); sel.push_back(q); }
        all.erase(all.begin() + long(j));
    }
}
static int check()
{
    int n_cases = 0;
    for (size_t m : {size_t(7), size_t(50), size_t(777), size_t(4000), size_t(11197)})
        for (double frac : {0.4, 0.97})
            for (int rep = 0; rep < 2; ++rep) {
                std::vector<uint8_t> valid; std::vector<double> J;
                make_rows(m, frac, unsigned(m) + 17u * unsigned(rep), rep == 1, valid, J);
                Rows R; R.valid = valid.data(); R.J = J.data(); R.m = m;
                for (double ratio : {0.1, 0.2, 0.5})
                    for (unsigned seed : {1u, 2u, 3u}) {
                        const size_t n_use = size_t(m * ratio);
                        auto h0 = [](double *H) { for (int i = 0; i < 36; ++i) H[i] = (i % 7 == 0) ? 1e-6 : 0.0; };
                        {   // random
                            std::mt19937 a(seed), b(seed); std::vector<size_t> sa, sb; double Ha[36], Hb[36]; h0(Ha); h0(Hb);
                            select_rnd(R, n_use, a, sa, Ha); rnd_literal(R, n_use, b, sb, Hb);
                            if (sa != sb || std::memcmp(Ha, Hb, sizeof(Ha)) || a() != b()) { std::printf("rnd differs: m %zu frac %g ratio %g seed %u\n", m, frac, ratio, seed); return 1; }
                        }
                        {   // greedy: lemma scoring against literal scoring
                            std::mt19937 a(seed), b(seed); std::vector<size_t> sa, sb; double Ha[36], Hb[36]; h0(Ha); h0(Hb);
                            unsetenv("MLH_SELECT_EXACT"); select_greedy(R, n_use, a, sa, Ha);
                            setenv("MLH_SELECT_EXACT", "1", 1); select_greedy(R, n_use, b, sb, Hb); unsetenv("MLH_SELECT_EXACT");
                            if (sa != sb || std::memcmp(Ha, Hb, sizeof(Ha)) || a() != b()) { std::printf("greedy differs: m %zu frac %g ratio %g seed %u repeats %d\n", m, frac, ratio, seed, rep); return 1; }
                        }
                        ++n_cases;
                    }
            }
    std::printf("ok %d cases\n", n_cases);
    return 0;
}
int main(int argc, char **argv){
  if (argc > 1 && std::string(argv[1]) == "--check") return check();
  for (int kind = 0; kind < 2; ++kind) {
    const size_t m = kind ? 11197 : 11732; const double frac = kind ? 0.40 : 0.97;
    std::vector<uint8_t> valid(m); std::vector<double> J(6*m);
    std::mt19937 g(5+kind); std::uniform_real_distribution<double> u(-1,1);
    for(size_t i=0;i<m;++i){ valid[i] = (u(g)*0.5+0.5) < frac; double n[3]={u(g),u(g),u(g)}; double nn=std::sqrt(n[0]*n[0]+n[1]*n[1]+n[2]*n[2]); for(double&x:n)x/=nn; double p[3]={30*u(g),30*u(g),3*u(g)}; double w=1.0+u(g)*0.5;
      J[6*i+0]=w*n[0];J[6*i+1]=w*n[1];J[6*i+2]=w*n[2]; J[6*i+3]=w*(p[1]*n[2]-p[2]*n[1]); J[6*i+4]=w*(p[2]*n[0]-p[0]*n[2]); J[6*i+5]=w*(p[0]*n[1]-p[1]*n[0]); }
    Rows R; R.valid=valid.data(); R.J=J.data(); R.m=m;
    for (int method = 0; method < 2; ++method) {
      double best=1e9; size_t chk=0, npick=0;
      for(int rep=0;rep<50;++rep){ std::mt19937 rng(7); std::vector<size_t> sel; double H[36]; for(int i=0;i<36;++i)H[i]=(i%7==0)?1e-6:0.0;
        auto t0=std::chrono::steady_clock::now();
        if(method==0) select_rnd(R,size_t(m*0.2),rng,sel,H); else select_greedy(R,size_t(m*0.2),rng,sel,H);
        auto t1=std::chrono::steady_clock::now(); double us=std::chrono::duration<double,std::micro>(t1-t0).count(); if(us<best)best=us; chk=0; for(size_t s:sel)chk=chk*1000003+s; npick=sel.size(); }
      printf("kind %d %s: %.0f us  picks %zu  checksum %zx\n",kind,method?"greedy":"rnd",best,npick,chk);
    }
  }
}
