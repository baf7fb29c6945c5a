#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <random>
#include "../../include/rt_abi.h"
int main(int argc,char**argv){
  std::mt19937_64 rng(atoll(argv[1])); int iters=atoi(argv[2]);
  std::vector<std::string> base;
  for(int i=3;i<argc;++i){ FILE* f=fopen(argv[i],"rb"); std::string s; char buf[65536]; size_t n; while((n=fread(buf,1,sizeof buf,f))>0) s.append(buf,n); fclose(f); base.push_back(s);}  
  const char* toks[]={"{","}","[","]","\"",":",",","1e999","-","null","99999999999999999999999","\\u12","\"width\":-1,","{\"Light\":{}}","0.0","-0.0","1e-400","true","\"Texture\"","\"radius\":0","nan","18446744073709551616","4294967296","\"samples_per_pixel\":0,"};
  int ok=0,err=0; char* jb=(char*)malloc(1<<22);
  for(int it=0;it<iters;++it){
    std::string d=base[rng()%base.size()];
    int nm=1+rng()%5;
    for(int m=0;m<nm && !d.empty();++m){
      size_t i=rng()%d.size();
      switch(rng()%5){
        case 0: d[i]="{}[]\",:0123456789-.eE \\"[rng()%24]; break;
        case 1: d.erase(i, 1+rng()%40); break;
        case 2: d.insert(i, toks[rng()%(sizeof toks/sizeof *toks)]); break;
        case 3: d.insert(i, d.substr(i>30?i-30:0, 30)); break;
        default: { // replace a number
          size_t j=d.find_first_of("0123456789", i); if(j!=std::string::npos){ size_t k=d.find_first_not_of("0123456789.eE-+", j); const char* nums[]={"0","-1","1e308","1e-320","4294967295","4294967296","65536","3","0.5","-0.0"}; d.replace(j, (k==std::string::npos?d.size():k)-j, nums[rng()%10]); } } break;
      }
    }
    char* p=(char*)malloc(d.size()+1); memcpy(p,d.data(),d.size()); p[d.size()]=0;  // (API takes text + len)
    RtSceneFile* sf=nullptr;
    int rc=rt_scene_load_string(p,d.size(),&sf);
    if(rc==0){ ok++; size_t need=0; rt_scene_to_json(sf,jb,1<<22,&need); double cam[11]; rt_scene_camera(sf,cam); const RtScene* s=rt_scene_get(sf); volatile uint32_t x=s->n_spheres; (void)x; rt_scene_free(sf);} else err++;
    free(p);
  }
  free(jb);
  printf("ok %d err %d\n",ok,err); return 0; }
