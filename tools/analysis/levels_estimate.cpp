// levels_estimate.cpp — development analysis (not product; round 5): would K uniform grids, one per RADIUS CLASS (bands of `ratio`),
// walked one after the other with a shared `closest`, beat the single grid + `large` list on worlds whose radii span decades?
// Rays: a pinhole camera like scenes/procedural.py's and up to five diffuse bounces.  Answer: no —
//   loguniform radii 0.05 .. 5: single 15.1 tests + 2.5 steps per ray; four levels 13.1 tests + 10.9 steps
//   bimodal 95 % r = 0.05 + 5 % r = 3: single 13.6 + 3.1; two levels 13.9 + 6.1;  BASELINE's uniform radii: 6.0 + 2.0 vs 4.1 + 3.4
// (every closest hit equal in both).  The tests are grazing rays crossing many cells of ANY grid, not the radius spread.
//   python: scenes/procedural.make_world(...) -> centre x, y, z, radius as f64 rows in a file;  g++ -O2 -std=c++17 -ffp-contract=off -mfma
//   -Iinclude tools/analysis/levels_estimate.cpp -o /tmp/lv;  /tmp/lv world.bin [ratio 4] [cells per sphere 2]
#include "../../rust-raytracer_amd/csrc/hip/rt_tables.h"
#include <random>
using namespace rtc;
struct Level { std::vector<RtSphere> sp; std::vector<uint32_t> map; HostTables t; DevScene ds; RtScene sc; };
static void finish(Level& L){
  memset(&L.sc,0,sizeof L.sc); L.sc.abi_version=RT_ABI_VERSION; L.sc.width=L.sc.height=16; L.sc.samples_per_pixel=1; L.sc.max_depth=5; L.sc.n_spheres=(uint32_t)L.sp.size(); L.sc.spheres=L.sp.data();
}
static void bind(Level& L){ fill_dev_scene(L.sc,L.t,L.ds); L.ds.geom=L.t.geom.data(); L.ds.matc=L.t.matc.data(); L.ds.cell_word=L.t.cell_word.data();
  L.ds.cell_items=L.t.grid.wide?reinterpret_cast<const uint16_t*>(L.t.cell_items32.data()):L.t.cell_items.data(); L.ds.large=L.t.large.data(); L.ds.large_geom=L.t.large_geom.data(); }
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); std::vector<double> raw; double b[4]; while(fread(b,8,4,f)==4){raw.insert(raw.end(),b,b+4);} fclose(f);
  const size_t n=raw.size()/4; double ratio=argc>2?atof(argv[2]):4.0; double cps=argc>3?atof(argv[3]):2.0;
  std::vector<RtSphere> all(n); memset(all.data(),0,sizeof(RtSphere)*n);
  for(size_t i=0;i<n;i++){ all[i].center[0]=raw[4*i];all[i].center[1]=raw[4*i+1];all[i].center[2]=raw[4*i+2];all[i].radius=raw[4*i+3]; }
  // single level: the product
  Level S; S.sp=all; finish(S); build_tables(S.sc,S.t); bind(S);
  printf("single: grid %ux%ux%u items %u large %u\n",S.t.grid.n[0],S.t.grid.n[1],S.t.grid.n[2],S.t.grid.n_items,S.t.grid.n_large);
  // classes: ground (r>=100) apart; others by radius bands of `ratio`
  double rmin=1e300; for(auto&s:all) if(fabs(s.radius)<100) rmin=std::min(rmin,fabs(s.radius));
  std::vector<Level> lv; Level G;  // G: the always-tested ones
  for(size_t i=0;i<n;i++){ double r=fabs(all[i].radius); if(r>=100){G.sp.push_back(all[i]);G.map.push_back(i);continue;}
    int k=(int)floor(log(r/rmin)/log(ratio)+1e-9); if((int)lv.size()<=k) lv.resize(k+1); lv[k].sp.push_back(all[i]); lv[k].map.push_back(i); }
  lv.erase(std::remove_if(lv.begin(),lv.end(),[](const Level&L){return L.sp.empty();}),lv.end());
  std::reverse(lv.begin(),lv.end());  // biggest spheres first: they bound `closest` for the finer levels
  for(auto&L:lv){ finish(L); std::string e=build_tables(L.sc,L.t);  // tables (geom...) ; then rebuild the grid without a `large` list
    GridParams gp; gp.cells_per_sphere=cps; gp.max_large_by_radius=0; gp.min_spheres=1; gp.large_cell_limit=1u<<30; build_grid(L.sc,L.t,gp);
    L.t.large_geom.resize(L.t.large.size()); for(size_t i=0;i<L.t.large.size();i++) L.t.large_geom[i]=L.t.geom[L.t.large[i]];
    bind(L); printf("  level: %zu spheres r in [%.3g..] grid %ux%ux%u items %u large %u\n",L.sp.size(),fabs(L.sp[0].radius),L.t.grid.n[0],L.t.grid.n[1],L.t.grid.n[2],L.t.grid.n_items,L.t.grid.n_large); }
  std::mt19937_64 g(7); std::uniform_real_distribution<double> U(-1,1);
  auto rnd_unit=[&](){ for(;;){V3 p=v3(U(g),U(g),U(g)); double l=length_squared(p); if(l<1&&l>1e-6) return muls(p,1/sqrt(l));} };
  double sT=0,sS=0,mT=0,mS=0; long rays=0, mism=0; double mGeoT=0; std::vector<long> histS(64,0),histM(64,0);
  V3 cam=v3(13,2,3), fwd=unit_vector(sub(v3(0,0,0),cam)); V3 right=unit_vector(v3(fwd.z,0,-fwd.x)); V3 up=v3(right.y*fwd.z-right.z*fwd.y, right.z*fwd.x-right.x*fwd.z, right.x*fwd.y-right.y*fwd.x);
  const double th=tan(10.0*M_PI/180.0);
  for(int s=0;s<300000;s++){
    V3 o=cam; V3 d=add(fwd, add(muls(right,U(g)*th*16.0/9.0), muls(up,U(g)*th)));
    for(int depth=0;depth<6;depth++){
      double c1=T_MAX; int b1=-1; uint32_t ne=0,ns=0; const GlobalTables tb{S.ds.geom,S.ds.matc}; hit_world_grid(S.ds,tb,o,d,c1,b1,ne,ns);
      double c2=T_MAX; int b2=-1; uint32_t me=0,ms=0; const RayK a=ray_consts(d);
      for(size_t i=0;i<G.sp.size();i++){ SphereGeom gg{G.sp[i].center[0],G.sp[i].center[1],G.sp[i].center[2],G.sp[i].radius}; me++; int bb=b2; if(exact_hit_any_order(o,d,a,gg,(uint32_t)G.map[i],c2,bb)) b2=bb; }
      for(auto&L:lv){ int bl=-1; double cl=c2; const GlobalTables tl{L.ds.geom,L.ds.matc}; uint32_t e0=0,s0=0; hit_world_grid(L.ds,tl,o,d,cl,bl,e0,s0); me+=e0; ms+=s0;
        if(bl>=0 && (cl<c2 || (cl==c2 && (int)L.map[bl]<b2))){ c2=cl; b2=(int)L.map[bl]; } }
      sT+=ne; sS+=ns; mT+=me; mS+=ms; rays++; histS[std::min<uint32_t>(63,ne+ns)]++; histM[std::min<uint32_t>(63,me+ms)]++;
      if(b1!=b2 || (b1>=0&&c1!=c2)) mism++;
      if(b1<0) break;
      V3 p=add(o,muls(d,c1)); const RtSphere& sp=all[b1]; V3 nrm=muls(sub(p,v3(sp.center[0],sp.center[1],sp.center[2])),1.0/sp.radius); if(dot(nrm,d)>0) nrm=neg(nrm);
      o=p; d=add(nrm,rnd_unit());
    }
  }
  printf("rays %ld mismatches %ld | single: tests %.2f steps %.2f | levels: tests %.2f steps %.2f\n",rays,mism,sT/rays,sS/rays,mT/rays,mS/rays);
  // wave-level proxy: max over 64 consecutive rays of (tests+steps)
  return 0; }
