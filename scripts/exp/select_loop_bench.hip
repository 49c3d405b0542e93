// Host-only timing of the good-feature selection loops (m-loam_amd/csrc/select.hip: select_rnd, select_greedy) on synthetic rows shaped like config 5's
// (11.7 k surf rows 97 % matched, 11.2 k corner rows 40 % matched, ratio 0.2). The translation unit is select.hip itself, so the loops timed are the
// product's; nothing here touches a GPU (the HIP symbols the file's launch helpers reference are left unresolved at link time and never called).
// Build + run: scripts/exp/select_loop_bench.sh. The checksum is over the selected indices: it must not move when the loops are changed.
#include "../../m-loam_amd/csrc/select.hip"
#include <cstdio>
using namespace mlh;
int main(int argc, char **argv){
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
