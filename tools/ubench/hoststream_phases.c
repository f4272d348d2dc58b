/* Phase timing of csrc/hoststream.c on the host: sequential MT19937 blocks, tempering, and the whole
 * standard_normal call for several thread counts.
 *   gcc -O3 -fopenmp -ffp-contract=off -o /tmp/hsp tools/ubench/hoststream_phases.c -lm && /tmp/hsp */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <stdint.h>
#include "../../physicsbasedbayesianinference_amd/csrc/hoststream.c"
static double now(){struct timespec t; clock_gettime(CLOCK_MONOTONIC,&t); return t.tv_sec+1e-9*t.tv_nsec;}
int main(){
  hs_state st; memset(&st,0,sizeof st); for(int i=0;i<624;i++) st.key[i]=1812433253u*i+12345u; st.pos=624;
  int64_t n=128*65536; double* out=malloc(n*8);
  for(int rep=0;rep<2;rep++){
    hs_stream s; stream_init(&s,&st);
    double t0=now(); size_t need=(size_t)(n/2/0.785*1.03)*4;
    size_t have=s.n_blocks*MT_N - s.pos0; size_t more=(need-have+MT_N-1)/MT_N;
    s.cap_blocks=s.n_blocks+more+1; s.blocks=realloc(s.blocks,s.cap_blocks*MT_N*4);
    double t1=now();
    for(size_t b=0;b<more;b++){ mt_next_block(s.blocks+(s.n_blocks-1)*MT_N, s.blocks+s.n_blocks*MT_N); ++s.n_blocks; }
    double t2=now();
    stream_ensure(&s,need);
    double t3=now();
    printf("alloc %.1f ms, mt blocks %.1f ms (%.2f ns/word), temper+alloc %.1f ms\n",(t1-t0)*1e3,(t2-t1)*1e3,(t2-t1)*1e9/(more*624.0),(t3-t2)*1e3);
    stream_free(&s);
  }
  if (getenv("HS_GEN")) pbbi_host_debug_set_gen(atoi(getenv("HS_GEN")), 8192, 1024);   /* generator threads (jump-ahead) */
  int ths[]={1,2,4,8,12,16,24,32,64};
  for(int k=0;k<9;k++){
    pbbi_host_set_threads(ths[k]);
    double best=1e9;
    for(int rep=0;rep<4;rep++){ double t4=now(); pbbi_host_standard_normal(&st,out,n); double t5=now(); if(t5-t4<best) best=t5-t4; }
    printf("threads %2d: full call best of 4 %.1f ms\n",ths[k],best*1e3);
#ifdef HS_PROFILE
    if(ths[k]==8||ths[k]==16){ for(int t=0;t<ths[k];t++){ printf("  t%02d gen %.1f waitblk %.1f count %.1f waitpref %.1f write %.1f chunks %.0f (ms, sum of 4 calls)\n",t,hs_prof[t][0]*1e3,hs_prof[t][1]*1e3,hs_prof[t][2]*1e3,hs_prof[t][3]*1e3,hs_prof[t][4]*1e3,hs_prof[t][5]); } }
    memset(hs_prof,0,sizeof hs_prof);
#endif
  }
  return 0;
}
