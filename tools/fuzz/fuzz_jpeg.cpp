#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <random>
#include "../../include/rt_abi.h"
extern "C" void rt_free(void*);
static std::vector<uint8_t> slurp(const char* p){ FILE* f=fopen(p,"rb"); std::vector<uint8_t> v; if(!f) return v; fseek(f,0,SEEK_END); long n=ftell(f); fseek(f,0,SEEK_SET); v.resize(n); fread(v.data(),1,n,f); fclose(f); return v; }
int main(int argc,char**argv){
  std::mt19937_64 rng(atoll(argv[1])); int iters=atoi(argv[2]);
  std::vector<std::vector<uint8_t>> base; for(int i=3;i<argc;++i) base.push_back(slurp(argv[i]));
  int ok=0,err=0;
  for(int it=0;it<iters;++it){
    std::vector<uint8_t> d=base[rng()%base.size()];
    // keep files small: cut the entropy-coded data to keep each decode fast
    if(d.size()>60000 && (rng()&3)) d.resize(20000+rng()%40000);
    int nm=1+rng()%8;
    for(int m=0;m<nm && !d.empty();++m){
      size_t hdr=d.size()<1500?d.size():1500; size_t i=(rng()%10<7)?rng()%hdr:rng()%d.size();
      switch(rng()%4){
        case 0: d[i]=(uint8_t)rng(); break;
        case 1: d.erase(d.begin()+i, d.begin()+std::min(d.size(), i+1+rng()%64)); break;
        case 2: { size_t k=1+rng()%16; std::vector<uint8_t> ins(k); for(auto&b:ins) b=(uint8_t)rng(); d.insert(d.begin()+i, ins.begin(), ins.end()); } break;
        default: d.resize(rng()%d.size()); break;
      }
    }
    // exact-size heap copy so that ASan sees any over-read
    uint8_t* p=(uint8_t*)malloc(d.size()?d.size():1); for(size_t k=0;k<d.size();++k)p[k]=d[k];
    uint8_t* rgb=nullptr; uint32_t w=0,h=0;
    int rc=rt_jpeg_decode_mem(p,d.size(),&rgb,&w,&h);
    if(rc==0){ ok++; volatile uint8_t x=rgb[(size_t)w*h*3-1]; (void)x; free(rgb);} else err++;
    free(p);
  }
  printf("ok %d err %d\n",ok,err); return 0; }
