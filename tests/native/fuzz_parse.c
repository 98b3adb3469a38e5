/* fuzz_parse.c -- test harness (built with -fsanitize=address,undefined by tests/test_parser_fuzz.py): mutated
 * codestreams through the host parser j2k_parse(), the frame splitter htj2k_splitter_* and the MXF essence walker
 * htj2k_mxf_next_essence.  Every packet is copied into an exact-size heap buffer, so an
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
        volatile uint64_t acc=0; for(int i=0;i<pl->nblocks;i++){ const J2kBlock*bk=pl->blocks+i; size_t e=(size_t)bk->data_off+bk->lcup+bk->lref;
          if(bk->flags & J2K_BLK_PART1){                    /* bytes + 0xFF 0xFF + trailer with the segment starts */
            const J2kPart1Trailer*tr=(const J2kPart1Trailer*)(pl->bytes+bk->data_off+J2K_P1_TRAILER_OFF(bk->lcup));
            e=(size_t)bk->data_off+J2K_P1_TRAILER_OFF(bk->lcup)+4+2*(size_t)bk->lref;
            if(e>pl->nbytes){printf("Part-1 block %d overruns pool\n",i);abort();}
            if(tr->nterm!=bk->lref||pl->bytes[bk->data_off+bk->lcup]!=0xFF||pl->bytes[bk->data_off+bk->lcup+1]!=0xFF){printf("Part-1 block %d: bad trailer\n",i);abort();}
            for(int k=0;k<tr->nterm;k++) if(tr->start[k]>bk->lcup){printf("Part-1 block %d: segment start outside\n",i);abort();}
            e=(size_t)bk->data_off+bk->lcup+2; } if(e>pl->nbytes){printf("block %d overruns pool\n",i);abort();} if(bk->lcup+bk->lref) acc+=pl->bytes[e-1]; if((size_t)bk->plane_off + (size_t)(bk->h-1)*bk->stride + bk->w > pl->nsamples){printf("block %d outside planes\n",i);abort();} }
      }
      /* the same bytes, twice over, through the frame splitter in random pieces (exact-size buffers again) */
      { htj2k_splitter*sp=NULL; if(htj2k_splitter_open(&sp)<0) abort();
        long pos=0, tot=2*m; int frames=0;
        while(pos<tot){ long k=1+rnd()%(tot-pos<4096?tot-pos:4096); uint8_t*q=malloc(k); for(long t=0;t<k;t++) q[t]=c[(pos+t)%(m?m:1)];
          long off=0; while(off<k){ const uint8_t*fr=NULL; int fs=0; int used=htj2k_splitter_parse(sp,q+off,(int)(k-off),&fr,&fs); if(used<0) break;
            if(fr&&fs>0){ volatile uint8_t t0=fr[0], t1=fr[fs-1]; (void)t0;(void)t1; frames++; } off+=used; if(!used&&!fr) break; }
          free(q); pos+=k; }
        { const uint8_t*fr=NULL; int fs=0; htj2k_splitter_parse(sp,NULL,0,&fr,&fs); if(fr&&fs>0){ volatile uint8_t t1=fr[fs-1]; (void)t1; } }
        htj2k_splitter_close(sp); (void)frames; }
      /* the same bytes behind an MXF essence key with a random BER length form, header bytes mutated, through the
       * KLV walker: every element handed out must lie inside the (exact-size) buffer */
      { static const uint8_t key[16]={0x06,0x0e,0x2b,0x34,0x01,0x02,0x01,0x01,0x0d,0x01,0x03,0x01,0x15,0x01,0x08,0x01};
        int nb=1+rnd()%8; long tot=3+16+1+nb+m+(rnd()%3?16:0); uint8_t*q=malloc(tot); long w=0;
        q[w++]=0x06;q[w++]=0x0e;q[w++]=0x2b; memcpy(q+w,key,16); w+=16; q[w++]=0x80|nb;
        for(int t=nb-1;t>=0;t--) q[w++]= t<8 ? (uint8_t)((uint64_t)m>>(8*t)) : 0;
        memcpy(q+w,c,m); w+=m; while(w<tot){ q[w]=key[w&15]; w++; }
        if(rnd()%2){ int k=1+rnd()%3; for(int i=0;i<k;i++) q[rnd()%(20+nb)]=rnd(); }
        if(rnd()%4==0) tot=rnd()%(tot+1);
        { uint8_t*e=malloc(tot+1); memcpy(e,q,tot); size_t pos=0; htj2k_mxf_essence es; int r2, guard=0;
          while((r2=htj2k_mxf_next_essence(e,(size_t)tot,&pos,&es))==1 && guard++<1000){
            if(es.data<e||es.data+es.size>e+tot||pos>(size_t)tot){printf("MXF element outside the buffer\n");abort();}
            if(es.size){ volatile uint8_t t0=es.data[0],t1=es.data[es.size-1]; (void)t0;(void)t1; } }
          free(e); }
        free(q); }
      free(c);
    }
    free(b);
  }
  printf("fuzz: %ld parses, %ld accepted\n",total,ok); j2k_parser_free(p); return 0; }
