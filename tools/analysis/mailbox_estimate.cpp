// mailbox_estimate.cpp — development analysis (not product; round 5): how many gridded tests of a segment repeat a sphere already
// tested in that segment (a sphere is listed in every cell it overlaps; the walk remembers only the LAST one tested), and what a
// mailbox of M entries would skip — per ray and as the maximum over groups of 64 rays (the wave statistic).  Camera rays + diffuse
// bounces on scenes/procedural.py worlds (centre x, y, z, radius as f64 rows in a file).  Result: BASELINE's uniform radii 2.01 ->
// 1.83 tests per ray with two entries (max over 64: 9.9 -> 8.2); log-uniform radii 7.09 -> 6.58 (22.1 -> 20.9), a perfect mailbox 5.60
// (14.9).  NOT built: in the kernel a skipped candidate still takes its lane a test round (the skip sits inside the round), so the
// wave's round count only falls if the candidate is dropped when the lane LANDS in the cell — round 3's `dedupe on landing`, measured
// +1.9 % (docs/history.md §4.5).
//   g++ -O2 -std=c++17 -ffp-contract=off -mfma -Iinclude tools/analysis/mailbox_estimate.cpp -o /tmp/mb; /tmp/mb world.bin
#include "../../rust-raytracer_amd/csrc/hip/rt_tables.h"
#include <random>
#include <set>
using namespace rtc;
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); std::vector<double> raw; double b[4]; while(fread(b,8,4,f)==4){raw.insert(raw.end(),b,b+4);} fclose(f);
  const size_t n=raw.size()/4;
  std::vector<RtSphere> all(n); memset(all.data(),0,sizeof(RtSphere)*n);
  for(size_t i=0;i<n;i++){ all[i].center[0]=raw[4*i];all[i].center[1]=raw[4*i+1];all[i].center[2]=raw[4*i+2];all[i].radius=raw[4*i+3]; }
  RtScene sc; memset(&sc,0,sizeof sc); sc.abi_version=RT_ABI_VERSION; sc.width=sc.height=16; sc.samples_per_pixel=1; sc.max_depth=5; sc.n_spheres=(uint32_t)n; sc.spheres=all.data();
  HostTables t; build_tables(sc,t);
  if(t.grid.wide){ std::fprintf(stderr,"mailbox_estimate: this world builds a WIDE grid (more than 65 535 spheres, > 4 095 items in a cell or >= 2^20 items): the tool decodes packed two-word cells only\n"); return 2; }
  DevScene ds; fill_dev_scene(sc,t,ds); ds.geom=t.geom.data(); ds.matc=t.matc.data(); ds.cell_word=t.cell_word.data(); ds.cell_items=t.cell_items.data(); ds.large=t.large.data(); ds.large_geom=t.large_geom.data();
  const GridDesc&G=t.grid; printf("grid %ux%ux%u items %u large %u\n",G.n[0],G.n[1],G.n[2],G.n_items,G.n_large);
  std::mt19937_64 g(7); std::uniform_real_distribution<double> U(-1,1);
  auto rnd_unit=[&](){ for(;;){V3 p=v3(U(g),U(g),U(g)); double l=length_squared(p); if(l<1&&l>1e-6) return muls(p,1/sqrt(l));} };
  V3 cam=v3(13,2,3), fwd=unit_vector(sub(v3(0,0,0),cam)); V3 right=unit_vector(v3(fwd.z,0,-fwd.x)); V3 up=v3(right.y*fwd.z-right.z*fwd.y, right.z*fwd.x-right.x*fwd.z, right.x*fwd.y-right.y*fwd.x);
  const double th=tan(10.0*M_PI/180.0);
  double tests[9]={0}; long rays=0; // tests[M] = gridded tests with a mailbox of M entries (0: none, 1: the product, 8: perfect)
  double maxsum[9]={0}; long groups=0; uint32_t gmax[9]={0}; int ingroup=0;
  for(int s=0;s<200000;s++){
    V3 o=cam; V3 d=add(fwd, add(muls(right,U(g)*th*16.0/9.0), muls(up,U(g)*th)));
    for(int depth=0;depth<6;depth++){
      const RayK a=ray_consts(d); double closest=T_MAX; int best=-1;
      for(uint32_t i=0;i<G.n_large;i++) exact_hit_any_order(o,d,a,t.geom[t.large[i]],t.large[i],closest,best);
      GridWalk w; int mode=grid_begin(G,o,d,w); uint32_t cnt[9]={0};
      if(mode==GRID_WALK){
        std::vector<uint32_t> hist;  // order of tested (distinct consecutive) spheres
        for(;;){ const uint32_t word=ds.cell_word[2*w.lin]; if(word==CELL_EXIT)break; const uint32_t first=word&CELL_START_MASK,count=word>>CELL_COUNT_SHIFT;
          for(uint32_t k=0;k<count;k++){ const uint32_t idx=ds.cell_items[first+k];
            for(int M=0;M<=8;M++){ bool skip=false; if(M==8){ skip=std::find(hist.begin(),hist.end(),idx)!=hist.end(); } else { for(int j=0;j<M&&j<(int)hist.size();j++) if(hist[hist.size()-1-j]==idx){skip=true;break;} } if(!skip) cnt[M]++; }
            if(std::find(hist.begin(),hist.end(),idx)==hist.end()) hist.push_back(idx); else { hist.erase(std::find(hist.begin(),hist.end(),idx)); hist.push_back(idx); }
            exact_hit_any_order(o,d,a,t.geom[idx],idx,closest,best); }
          if(best>=0&&grid_done(w,closest))break; grid_step(w); } }
      for(int M=0;M<=8;M++){ tests[M]+=cnt[M]; gmax[M]=std::max(gmax[M],cnt[M]); } rays++;
      if(++ingroup==64){ for(int M=0;M<=8;M++){maxsum[M]+=gmax[M]; gmax[M]=0;} groups++; ingroup=0; }
      if(best<0)break; V3 p=add(o,muls(d,closest)); const RtSphere& sp=all[best]; V3 nrm=muls(sub(p,v3(sp.center[0],sp.center[1],sp.center[2])),1.0/sp.radius); if(dot(nrm,d)>0)nrm=neg(nrm); o=p; d=add(nrm,rnd_unit());
    } }
  printf("gridded tests per ray: no mailbox %.2f | 1 entry (product) %.2f | 2: %.2f | 3: %.2f | 4: %.2f | perfect %.2f\n",tests[0]/rays,tests[1]/rays,tests[2]/rays,tests[3]/rays,tests[4]/rays,tests[8]/rays);
  printf("max over groups of 64 rays:  no mailbox %.2f | 1 entry %.2f | 2: %.2f | 3: %.2f | 4: %.2f | perfect %.2f\n",maxsum[0]/groups,maxsum[1]/groups,maxsum[2]/groups,maxsum[3]/groups,maxsum[4]/groups,maxsum[8]/groups);
}
