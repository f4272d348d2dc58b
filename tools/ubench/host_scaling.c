/* Does the polar transform scale over the host's cores when memory is out of the picture?  Every thread
 * transforms its own L2-resident chunk repeatedly; prints the wall time per thread count and the CPUs the
 * threads ran on.   gcc -O3 -fopenmp -ffp-contract=off -o /tmp/hscale tools/ubench/host_scaling.c -lm */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <stdint.h>
#include <sched.h>
#include "../../physicsbasedbayesianinference_amd/csrc/hoststream.c"
static double now(){struct timespec t; clock_gettime(CLOCK_MONOTONIC,&t); return t.tv_sec+1e-9*t.tv_nsec;}
int main(){
  const int CH=1<<15, TOTAL=336;
  int ths[]={1,2,4,8,12,16,24,32};
  for(int k=0;k<8;k++){
    int T=ths[k]; int cpus[64]; double sums[64];
    double best=1e9;
    for(int rep=0;rep<3;rep++){
      double t0=now();
#pragma omp parallel num_threads(T)
      {
        int tid=omp_get_thread_num();
        uint32_t* w=malloc(CH*16); double* o=malloc(CH*16);
        uint32_t x=12345u+tid; for(int i=0;i<CH*4;i++){ x=x*1664525u+1013904223u; w[i]=x; }
        double acc=0;
        for(int c=tid;c<TOTAL;c+=T){
          int kk=0; for(int a=0;a<CH;a++){ double f,g; if(attempt(w+a*4,&f,&g)){ o[2*kk]=f; o[2*kk+1]=g; kk++; } }
          acc+=o[kk];
        }
        cpus[tid]=sched_getcpu(); sums[tid]=acc; free(w); free(o);
      }
      double t=now()-t0; if(t<best) best=t;
    }
    printf("threads %2d: %.1f ms; cpus:",T,best*1e3); for(int i=0;i<T;i++) printf(" %d",cpus[i]); printf("\n");
  }
  return 0;
}
