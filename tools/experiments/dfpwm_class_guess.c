#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { int cu, s, pb; } Enc;
static inline int med3(int a,int b,int c){int mn=a<b?a:b,mx=a<b?b:a; int t=mx<c?mx:c; return mn>t?mn:t;}
static inline int step(Enc*e, unsigned u){
  int lim = e->cu<254?e->cu:254; int bit = (int)u>lim; int target=bit?255:0,b=bit?1:-1; int diff=target-e->cu;
  int st=(e->s*diff+512)>>10; st=med3(st,b,diff); e->cu+=st; int ns=b*e->pb+e->s; e->s=med3(ns,8,1023); e->pb=b; return bit; }
static unsigned pack(Enc e){return e.cu|e.s<<8|(e.pb>0?1<<18:0);}
static int inv(Enc e,long t){ int q=e.pb>0?0:1; return (int)(((e.s-2*q-t)%4+4)%4); }
int main(int argc,char**argv){
  int C=atoi(argv[1]); int S0=atoi(argv[2]);
  int Ws[]={1024,1536,2048,2560,3072,4096}; long miss[6]={0},tot[6]={0};
  for(int a=3;a<argc;a++){
  FILE*f=fopen(argv[a],"rb"); static signed char buf[1<<20]; int n=fread(buf,1,sizeof buf,f); fclose(f);
  static unsigned truth[1<<20]; static unsigned char tinv[1<<20]; Enc e={128,0,-1};
  for(int i=0;i<n;i++){ truth[i]=pack(e); tinv[i]=inv(e,i); step(&e,(unsigned)(buf[i]+128)); }
  int changes=0; for(int i=2049;i<n;i++) if(tinv[i]!=tinv[2048]) {changes++;break;}
  for(int wi=0;wi<6;wi++){ int W=Ws[wi];
      for(int p=8192;p<n;p+=C){ tot[wi]++;
          Enc c={buf[p-W]+128,S0,-1}; long t0=p-W; int I=tinv[2048];
          while(inv(c,t0)!=I) c.s++;
          for(int i=p-W;i<p;i++)step(&c,(unsigned)(buf[i]+128));
          if(pack(c)!=truth[p]) miss[wi]++; } }
  printf("%s class-change-after-2048:%d\n",argv[a],changes);
  }
  for(int wi=0;wi<6;wi++)printf("W=%d miss %ld / %ld = %.5f\n",Ws[wi],miss[wi],tot[wi],(double)miss[wi]/tot[wi]);
}
