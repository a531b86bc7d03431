// touch every page of a file through mmap vs pread, N threads: where does the time of a parallel text parser go?
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
static const char *base; static size_t size; static int nthreads, fd, mode;
static double now(void){struct timespec t; clock_gettime(CLOCK_MONOTONIC,&t); return t.tv_sec+1e-9*t.tv_nsec;}
static void *work(void *arg){ long id=(long)arg; size_t a=size/nthreads*id, z=id==nthreads-1?size:size/nthreads*(id+1); unsigned long s=0;
  if(mode==0){ for(size_t i=a;i<z;i+=4096) s+=(unsigned char)base[i]; }
  else if(mode==1){ for(size_t i=a;i<z;i++) s+=(unsigned char)base[i]=='\n'; }
  else { char *buf=malloc(1<<20); for(size_t i=a;i<z;i+=1<<20){ size_t n=z-i<(1<<20)?z-i:(1<<20); ssize_t r=pread(fd,buf,n,i); for(ssize_t k=0;k<r;k++) s+=(unsigned char)buf[k]=='\n'; } free(buf);}
  return (void*)s; }
int main(int argc,char**argv){ fd=open(argv[1],O_RDONLY); struct stat st; fstat(fd,&st); size=st.st_size;
  for(int rep=0;rep<2;rep++) for(mode=0;mode<3;mode++) for(nthreads=16;nthreads<=64;nthreads*=4){
    base=mmap(NULL,size,PROT_READ,MAP_PRIVATE,fd,0); pthread_t th[64]; double t0=now();
    for(long i=0;i<nthreads;i++) pthread_create(&th[i],NULL,work,(void*)i); unsigned long tot=0; for(int i=0;i<nthreads;i++){void*r; pthread_join(th[i],&r); tot+=(unsigned long)r;}
    double dt=now()-t0; munmap((void*)base,size);
    printf("%s threads %d: %.3f s (%.1f GB/s) sum %lu\n", mode==0?"mmap touch 1 B/page":mode==1?"mmap count newlines":"pread 1 MB + count newlines", nthreads, dt, size/dt/1e9, tot); }
  return 0; }
