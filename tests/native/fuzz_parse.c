/* fuzz_parse.c -- test harness (built with -fsanitize=address,undefined by tests/test_parser_fuzz.py): mutated
 * codestreams through the host parser j2k_parse().  Every packet is copied into an exact-size heap buffer, so an
 * over-read of the packet is an ASan report; every accepted plan is checked the way the device layer would walk it
 * (block bytes inside the byte pool, block windows inside the coefficient planes).
 * usage: fuzz_parse ITERATIONS file... */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "htj2k_amd.h"
#include "j2k_plan.h"
static uint32_t rs = 12345;
static uint32_t rnd(void){ rs = rs*1664525u+1013904223u; return rs>>8; }
int main(int argc,char**argv){
  int iters = argc > 1 ? atoi(argv[1]) : 100;
  J2kParser*p=j2k_parser_new(); htj2k_opts o; memset(&o,0,sizeof o); o.req_pix_fmt=-1;
  long total=0, ok=0;
  for(int a=2;a<argc;a++){
    FILE*f=fopen(argv[a],"rb"); if(!f) continue; fseek(f,0,SEEK_END); long n=ftell(f); fseek(f,0,SEEK_SET);
    uint8_t*b=malloc(n); if(fread(b,1,n,f)!=(size_t)n){} fclose(f);
    for(int it=0;it<iters;it++){
      long m=n; int mode=rnd()%5;
      if(mode==0) m = rnd()%(n+1);                       /* truncation */
      uint8_t*c=malloc(m+1); memcpy(c,b,m);               /* exact-size buffer: ASan catches over-reads */
      if(mode>=1 && m>0){ int k=1+rnd()%(mode==4?64:4); for(int i=0;i<k;i++){ long pos=(mode==2)? rnd()%(m<200?m:200) : rnd()%m; if(mode==3) c[pos]^=1<<(rnd()%8); else c[pos]=rnd(); } }
      const J2kPlan*pl=NULL; o.reduction_factor = (it%7==0)? rnd()%4 : 0;
      int r=j2k_parse(p,c,(int)m,&o,0,&pl); total++; if(r>=0) { ok++;
        /* touch everything the device layer would read */
        volatile uint64_t acc=0; for(int i=0;i<pl->nblocks;i++){ const J2kBlock*bk=pl->blocks+i; size_t e=(size_t)bk->data_off+bk->lcup+bk->lref; if(e>pl->nbytes){printf("block %d overruns pool\n",i);abort();} if(bk->lcup+bk->lref) acc+=pl->bytes[e-1]; if((size_t)bk->plane_off + (size_t)(bk->h-1)*bk->stride + bk->w > pl->nsamples){printf("block %d outside planes\n",i);abort();} }
      }
      free(c);
    }
    free(b);
  }
  printf("fuzz: %ld parses, %ld accepted\n",total,ok); j2k_parser_free(p); return 0; }
