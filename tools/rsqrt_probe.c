// Characterise this CPU's rsqrtss: how many leading mantissa bits it depends on, and whether
// the result scales exactly with the exponent (parity model).
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static float rs(float x){float r;__asm__ __volatile__("movss %1, %%xmm0\n rsqrtss %%xmm0, %%xmm1\n movss %%xmm1, %0\n":"=m"(r):"m"(x):"xmm0","xmm1");return r;}
int main(){
  FILE*f=fopen("/proc/cpuinfo","r"); char line[256]; while(f&&fgets(line,256,f)) if(!strncmp(line,"model name",10)){printf("%s",line);break;} if(f)fclose(f);
  const uint32_t NM=1u<<23; uint32_t*out=malloc(2*(size_t)NM*4);
  for(uint32_t par=0;par<2;par++){uint32_t e=par?127:128; for(uint32_t m=0;m<NM;m++){uint32_t u=(e<<23)|m; float x,r; memcpy(&x,&u,4); r=rs(x); memcpy(&out[par*NM+m],&r,4);} }
  for(int k=0;k<=23;k++){uint32_t mask=~((1u<<(23-k))-1)&(NM-1); long bad=0; for(uint32_t par=0;par<2;par++) for(uint32_t m=0;m<NM;m++) if(out[par*NM+m]!=out[par*NM+(m&mask)]) bad++; printf("k=%2d mismatches=%ld\n",k,bad); if(!bad) break;}
  // number of distinct outputs and low mantissa bits of outputs
  uint32_t orr=0; long distinct=0; for(uint32_t par=0;par<2;par++) for(uint32_t m=0;m<NM;m++){orr|=out[par*NM+m]; if(m==0||out[par*NM+m]!=out[par*NM+m-1]) distinct++;}
  printf("distinct runs=%ld, OR of output bits=%08x\n",distinct,orr);
  // exponent scaling: compare e vs canonical
  long sbad=0,tot=0; for(uint32_t e=1;e<255;e++){ uint32_t par=e&1; int ec=par?127:128; int q=((int)e-ec)/2; for(uint32_t m=0;m<NM;m+=4099){uint32_t u=(e<<23)|m; float x,r; memcpy(&x,&u,4); r=rs(x); uint32_t rb; memcpy(&rb,&r,4); uint32_t want=out[par*NM+m]-(q<<23); tot++; if(rb!=want) sbad++; }}
  printf("exponent scaling mismatches=%ld of %ld\n",sbad,tot);
  // print a few samples
  for(int i=0;i<8;i++){uint32_t u=(127u<<23)|(i*0x123457u&0x7fffff); float x; memcpy(&x,&u,4); float r=rs(x); uint32_t rb; memcpy(&rb,&r,4); printf("x=%08x r=%08x\n",u,rb);}
  return 0;}
