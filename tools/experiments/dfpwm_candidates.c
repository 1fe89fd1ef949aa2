#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { int cu, s, pb; } Enc;
static inline int med3(int a,int b,int c){int mn=a<b?a:b,mx=a<b?b:a; int t=mx<c?mx:c; return mn>t?mn:t;}
static inline int step(Enc*e, unsigned u){
  int lim = e->cu<254?e->cu:254; int bit = (int)u>lim; int target=bit?255:0,b=bit?1:-1; int diff=target-e->cu;
  int st=(e->s*diff+512)>>10; st=med3(st,b,diff); e->cu+=st; int ns=b*e->pb+e->s; e->s=med3(ns,8,1023); e->pb=b; return bit; }
static unsigned pack(Enc e){return e.cu|e.s<<8|(e.pb>0?1<<18:0);}
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); static signed char buf[1<<20]; int n=fread(buf,1,sizeof buf,f); fclose(f);
  int C=atoi(argv[2]); // chunk
  static unsigned truth[1<<20]; Enc e={128,0,-1};
  long ones=0,same=0; int pbit=0; long shist[11]={0};
  for(int i=0;i<n;i++){ truth[i]=pack(e); int b=step(&e,(unsigned)(buf[i]+128)); ones+=b; same+=(b==pbit); pbit=b; int k=0; while((8<<k)<e.s&&k<10)k++; shist[k]++; }
  printf("%s n=%d same-frac=%.3f strength hist(<=8,16,32,...):",argv[1],n,(double)same/n); for(int k=0;k<11;k++)printf(" %.3f",(double)shist[k]/n); printf("\n");
  int Ws[]={256,512,1024,2048,4096,8192};
  for(int wi=0;wi<6;wi++){ int W=Ws[wi];
    int nb=0; double fracsum=0; int distinctsum=0, maxdist=0; int hit_guess[8]={0}; int guesses[8]={8,16,24,32,48,64,128,1023}; int hit_majority=0;
    for(int p=C;p+0<n;p+=C){ if(p<W) continue; nb++;
      static unsigned ends[2032]; 
      for(int id=0;id<2032;id++){ Enc c={buf[p-W]+128,8+(id>>1),(id&1)?1:-1}; for(int i=p-W;i<p;i++)step(&c,(unsigned)(buf[i]+128)); ends[id]=pack(c);} 
      int hit=0; for(int id=0;id<2032;id++)hit+=ends[id]==truth[p]; fracsum+=hit/2032.0;
      // distinct + majority
      static unsigned d[2032]; static int cnt[2032]; int nd=0; for(int id=0;id<2032;id++){int k;for(k=0;k<nd;k++)if(d[k]==ends[id]){cnt[k]++;break;} if(k==nd){d[nd]=ends[id];cnt[nd]=1;nd++;}}
      distinctsum+=nd; if(nd>maxdist)maxdist=nd; int bm=0; for(int k=1;k<nd;k++)if(cnt[k]>cnt[bm])bm=k; hit_majority+= d[bm]==truth[p];
      for(int g=0;g<8;g++){ int ok=0; for(int pbv=0;pbv<2;pbv++){ int id=(guesses[g]-8)*2+pbv; if(ends[id]==truth[p]) ok++; } hit_guess[g]+= ok; }
    }
    printf(" W=%5d chunks=%d mean frac of cands==truth %.3f, distinct mean %.1f max %d, majority==truth %.3f; guess s0 hit(both pb /2):",W,nb,fracsum/nb,(double)distinctsum/nb,maxdist,(double)hit_majority/nb);
    for(int g=0;g<8;g++)printf(" %d:%.2f",guesses[g],hit_guess[g]/(2.0*nb)); printf("\n");
  }
}
