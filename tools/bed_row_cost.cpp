// component costs of a BED row parse (single thread)
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <string>
#include <string_view>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
static double now(){ return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc,char**argv){
  int fd=open(argv[1],O_RDONLY); struct stat st; fstat(fd,&st); size_t n=st.st_size;
  const char*base=(const char*)mmap(0,n,PROT_READ,MAP_PRIVATE,fd,0);
  for(int rep=0;rep<2;rep++){
   { double t0=now(); size_t a=0,lines=0; while(a<n){ const char*nl=(const char*)memchr(base+a,'\n',n-a); size_t e=nl?nl-base:n; a=e+1; lines++; } printf("memchr only: %.1f ns/row\n",(now()-t0)*1e9/lines); }
   { double t0=now(); size_t a=0,lines=0; uint64_t acc=0; while(a<n){ const char*nl=(const char*)memchr(base+a,'\n',n-a); size_t e=nl?nl-base:n; const char*p=base+a,*q=p,*ee=base+e; uint64_t hi=0; for(;q+8<=ee;q+=8){uint64_t w; memcpy(&w,q,8); hi|=w;} for(;q<ee;++q) hi|=(unsigned char)*q; acc+=hi; a=e+1; lines++; } printf("+hi check: %.1f ns/row (%llu)\n",(now()-t0)*1e9/lines,(unsigned long long)acc); }
   { double t0=now(); size_t a=0,lines=0; uint64_t acc=0; while(a<n){ const char*nl=(const char*)memchr(base+a,'\n',n-a); size_t e=nl?nl-base:n; const char*q=base+a,*ee=base+e; 
        uint64_t h=1469598103934665603ull; while(q<ee&&*q!='\t'){h=(h^(unsigned char)*q)*1099511628211ull;++q;} ++q; uint32_t v1=0; while(q<ee){unsigned d=(unsigned char)*q-'0'; if(d>9)break; v1=v1*10+d;++q;} ++q; uint32_t v2=0; while(q<ee){unsigned d=(unsigned char)*q-'0'; if(d>9)break; v2=v2*10+d;++q;} acc+=h+v1+v2; a=e+1; lines++; } printf("memchr+hash+2 numbers: %.1f ns/row (%llu)\n",(now()-t0)*1e9/lines,(unsigned long long)acc); }
   { double t0=now(); size_t lines=0; uint64_t acc=0; const char*q=base,*end=base+n; while(q<end){ uint64_t h=1469598103934665603ull; while(*q!='\t'){h=(h^(unsigned char)*q)*1099511628211ull;++q;} ++q; uint32_t v1=0; for(;;){unsigned d=(unsigned char)*q-'0'; if(d>9)break; v1=v1*10+d;++q;} ++q; uint32_t v2=0; for(;;){unsigned d=(unsigned char)*q-'0'; if(d>9)break; v2=v2*10+d;++q;} while(q<end&&*q!='\n')++q; ++q; acc+=h+v1+v2; lines++; } printf("no memchr, one pass: %.1f ns/row (%llu)\n",(now()-t0)*1e9/lines,(unsigned long long)acc); }
  }
}
