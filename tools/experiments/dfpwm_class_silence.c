#include <stdio.h>
#include <stdlib.h>
typedef struct { int cu, s, pb; } Enc;
static inline int med3(int a,int b,int c){int mn=a<b?a:b,mx=a<b?b:a; int t=mx<c?mx:c; return mn>t?mn:t;}
static int clamped;
static inline int step(Enc*e, unsigned u){
  int lim = e->cu<254?e->cu:254; int bit = (int)u>lim; int target=bit?255:0,b=bit?1:-1; int diff=target-e->cu;
  int st=(e->s*diff+512)>>10; st=med3(st,b,diff); e->cu+=st; int ns=b*e->pb+e->s; e->s=med3(ns,8,1023); clamped=(e->s!=ns); e->pb=b; return bit; }
static unsigned pack(Enc e){return e.cu|e.s<<8|(e.pb>0?1<<18:0);}
static int inv(Enc e,long t){ int q=e.pb>0?0:1; return (int)(((e.s-2*q-t)%4+4)%4); }
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); static signed char buf[1<<20]; int n=fread(buf,1,sizeof buf,f); fclose(f);
  int C=atoi(argv[2]), W=atoi(argv[3]);
  static unsigned truth[1<<20]; static unsigned char tinv[1<<20]; Enc e={128,0,-1}; int nclamp=0; static int clampcount[1<<20];
  for(int i=0;i<n;i++){ truth[i]=pack(e); tinv[i]=inv(e,i); step(&e,(unsigned)(buf[i]+128)); nclamp+=clamped; clampcount[i]=nclamp; }
  printf("total clamps %d\n",nclamp);
  // per chunk boundary: true class, clamps in previous chunk, hit with true class, hit with class of previous boundary
  int prevI=tinv[512]; int nb=0,hit_true=0,hit_prev=0,changes=0, hit_any=0;
  for(int p=C;p<n;p+=C){ nb++; int I=tinv[p]; if(I!=prevI)changes++;
    int ok[4];
    for(int k=0;k<4;k++){ Enc c={buf[p-W]+128,40,-1}; while(inv(c,p-W)!=k)c.s++; for(int i=p-W;i<p;i++)step(&c,(unsigned)(buf[i]+128)); ok[k]=pack(c)==truth[p]; }
    hit_true+=ok[I]; hit_prev+=ok[prevI]; hit_any+= ok[0]|ok[1]|ok[2]|ok[3];
    if(nb<=40) printf("p=%d cls=%d clamps_in_chunk=%d ok[0..3]=%d%d%d%d\n",p,I,clampcount[p-1]-clampcount[p-C-1>0?p-C-1:0],ok[0],ok[1],ok[2],ok[3]);
    prevI=I; }
  printf("chunks %d class changes %d hit(true class) %d hit(prev boundary class) %d hit(any of 4) %d\n",nb,changes,hit_true,hit_prev,hit_any);
}
