// The host feed's DEFLATE decoder (kmerdb_amd/csrc/kdb_inflate.cpp.h) and gzip stream reader (GzStream) against zlib, under
// AddressSanitizer + UBSan on the CPU: every block type (stored / fixed / dynamic, every level and strategy), data of six kinds,
// arenas that are drained every 1 .. 2^20 bytes (resumption at any symbol), then single-bit corruptions and truncations -- the
// decoder must fail exactly when zlib does and never touch memory outside its arena.   Build + run: tests/test_host_sanitize.py
//   inflate_check ITERATIONS [speed]
#include "../../kmerdb_amd/csrc/kdb_hostparse.cpp.h"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <chrono>
#include <string>
#include <unistd.h>
using namespace kdbhost;
// raw inflate of `comp` with the new decoder, through an arena with small data part (exercises resumption); returns false on error
static bool mine(const std::vector<uint8_t>&comp, std::vector<uint8_t>&out, size_t data_cap, const char **err, size_t *consumed){
  Inflater *inf=new Inflater(); inf->reset(comp.data(), comp.data()+comp.size());
  std::vector<uint8_t> arena(Inflater::WINDOW+data_cap+512);
  uint8_t *o=arena.data()+Inflater::WINDOW; size_t hist=0; out.clear();
  while(true){
    uint8_t *base=arena.data()+Inflater::WINDOW;
    bool ok=inf->run(o, base+data_cap, base-hist);
    out.insert(out.end(), base, o);
    if(!ok){*err=inf->err; delete inf; return false;}
    size_t n=o-base; size_t tot=hist+n; size_t keep=tot<Inflater::WINDOW?tot:Inflater::WINDOW;
    memmove(base-keep, o-keep, keep); hist=keep; o=base;
    if(inf->state==Inflater::DONE) break;
  }
  *consumed=inf->input_position()-comp.data(); delete inf; return true;
}
static std::vector<uint8_t> zdeflate(const std::vector<uint8_t>&in,int level,int strategy){
  z_stream zs; memset(&zs,0,sizeof zs); deflateInit2(&zs,level,Z_DEFLATED,-15,8,strategy);
  std::vector<uint8_t> out(deflateBound(&zs,in.size())+64);
  zs.next_in=(Bytef*)in.data(); zs.avail_in=in.size(); zs.next_out=out.data(); zs.avail_out=out.size();
  deflate(&zs,Z_FINISH); out.resize(zs.total_out); deflateEnd(&zs); return out;
}
static int zinflate(const std::vector<uint8_t>&comp, std::vector<uint8_t>&out){
  z_stream zs; memset(&zs,0,sizeof zs); inflateInit2(&zs,-15); out.resize(1<<20); size_t tot=0;
  zs.next_in=(Bytef*)comp.data(); zs.avail_in=comp.size();
  int rc;
  do{ if(tot==out.size()) out.resize(out.size()*2); zs.next_out=out.data()+tot; zs.avail_out=out.size()-tot; rc=inflate(&zs,Z_NO_FLUSH); tot=out.size()-zs.avail_out; if(rc==Z_BUF_ERROR&&zs.avail_in==0) break;}while(rc==Z_OK);
  out.resize(tot); inflateEnd(&zs); return rc;
}
int main(int argc,char**argv){
  const int iters = argc>1? atoi(argv[1]) : 400; const bool speed = argc>2;
  std::mt19937_64 rng(7); int ncase=0, nbad=0;
  for(int it=0; it<iters; it++){
    size_t n = (it%7==0)? 0 : (size_t)(rng()% (it%5==0? 600000: 40000));
    std::vector<uint8_t> data(n);
    int kind=it%6;
    for(size_t i=0;i<n;i++){
      if(kind==0) data[i]=rng()&0xFF;                       // random: stored blocks
      else if(kind==1) data[i]="ACGT"[rng()&3];              // sequence
      else if(kind==2) data[i]=(i%151==150)?'\n':"ACGTN"[rng()%5];
      else if(kind==3) data[i]=(uint8_t)('A'+ (i/ (1+ (rng()%3)))%3); // repetitive
      else if(kind==4) data[i]= (i%300<150)? "ACGT"[rng()&3] : 'I'; // fastq-like
      else data[i]=(uint8_t)(rng()%7==0? rng()&0xFF : 'x');
    }
    int level = (int)(rng()%10); int strat = (it%11==0)? Z_FIXED : (it%13==0? Z_HUFFMAN_ONLY : (it%17==0? Z_RLE: Z_DEFAULT_STRATEGY));
    auto comp=zdeflate(data,level,strat);
    comp.push_back(0xAA); comp.push_back(0xBB);            // bytes after the stream (a gzip trailer follows in real life)
    std::vector<uint8_t> out; const char*err=""; size_t consumed=0;
    size_t cap = (size_t[]){1,7,300,4096,65536,1<<20}[rng()%6];
    bool ok=mine(comp,out,cap,&err,&consumed); ncase++;
    if(!ok||out!=data||consumed!=comp.size()-2){ printf("MISMATCH it=%d n=%zu level=%d strat=%d ok=%d err=%s out=%zu consumed=%zu of %zu\n",it,n,level,strat,ok,err,out.size(),consumed,comp.size()-2); nbad++; }
    // corrupt a byte / truncate: never crash; when zlib reports an error we must too, and if neither does the outputs agree
    for(int c=0;c<4;c++){
      auto bad=comp; bad.resize(bad.size()-2);
      if(bad.empty()) break;
      if(c%2==0) bad[rng()%bad.size()]^= (uint8_t)(1u<<(rng()%8)); else bad.resize(rng()%bad.size());
      std::vector<uint8_t> zo, mo; int zrc=zinflate(bad,zo); const char*e2=""; size_t c2=0;
      bool mok=mine(bad,mo,cap,&e2,&c2); ncase++;
      bool zok = zrc==Z_STREAM_END;
      if(zok!=mok || (zok && zo!=mo)){ printf("CORRUPT-DIFF it=%d c=%d zlib rc=%d mine ok=%d err=%s sizes %zu %zu\n",it,c,zrc,mok,e2,zo.size(),mo.size()); nbad++; }
    }
  }
  // the stream reader: two members + zero padding in one file, read back in odd-sized pieces; then a corrupted copy
  {
    std::vector<uint8_t> data(9u<<20); for(size_t i=0;i<data.size();i++) data[i]= (i%300<150)? "ACGT"[rng()&3] : 'I';
    char path[64]; snprintf(path,sizeof path,"/tmp/kdb_inflate_check_%d.gz",(int)getpid());
    for(int variant=0; variant<3; variant++){
      std::vector<uint8_t> file;
      size_t cut=data.size()/3;
      for(int m=0;m<2;m++){
        z_stream zs; memset(&zs,0,sizeof zs); deflateInit2(&zs,m?6:1,Z_DEFLATED,15+16,8,Z_DEFAULT_STRATEGY);
        const uint8_t *p=data.data()+(m?cut:0); size_t n=m?data.size()-cut:cut;
        std::vector<uint8_t> out(deflateBound(&zs,n)+64); zs.next_in=(Bytef*)p; zs.avail_in=n; zs.next_out=out.data(); zs.avail_out=out.size();
        deflate(&zs,Z_FINISH); out.resize(zs.total_out); deflateEnd(&zs); file.insert(file.end(),out.begin(),out.end());
        if(m==0) file.insert(file.end(),5,0);
      }
      if(variant==1) file[file.size()/2]^=0x10;
      if(variant==2) file.resize(file.size()-20);
      FILE*f=fopen(path,"wb"); fwrite(file.data(),1,file.size(),f); fclose(f);
      const char*why=""; GzStream*g=gz_open(path,&why); ncase++;
      if(!g){ printf("gz_open failed: %s\n",why); nbad++; continue; }
      std::vector<uint8_t> back; std::vector<uint8_t> piece(3000017); bool failed=false;
      for(;;){ size_t n=0; if(g->read(piece.data(), 1+rng()%piece.size(), &n)){ failed=true; break;} if(!n) break; back.insert(back.end(),piece.begin(),piece.begin()+n); }
      if(variant==0 && (failed || back!=data)){ printf("GzStream round trip failed (%s) %zu vs %zu\n", g->err.c_str(), back.size(), data.size()); nbad++; }
      if(variant!=0 && !failed){ printf("GzStream accepted a damaged file (variant %d)\n",variant); nbad++; }
      delete g;
    }
    remove(path);
  }
  printf("%d cases, %d bad\n",ncase,nbad);
  if(!speed) return nbad!=0;
  // speed on FASTQ-like text
  std::vector<uint8_t> data; data.reserve(200<<20); int i=0;
  while(data.size()<(150u<<20)){ char h[32]; int hl=snprintf(h,32,"@r%d\n",i++); data.insert(data.end(),h,h+hl); for(int b=0;b<150;b++) data.push_back("ACGT"[rng()&3]); data.push_back('\n'); data.push_back('+'); data.push_back('\n'); data.insert(data.end(),150,'I'); data.push_back('\n'); }
  for(int level: {1,6}){
    auto comp=zdeflate(data,level,Z_DEFAULT_STRATEGY); std::vector<uint8_t> out; const char*err=""; size_t consumed;
    bool ok=mine(comp,out,4<<20,&err,&consumed);
    auto t0=std::chrono::steady_clock::now();
    { Inflater *inf=new Inflater(); inf->reset(comp.data(), comp.data()+comp.size()); size_t cap=4<<20; std::vector<uint8_t> arena(Inflater::WINDOW+cap+512);
      uint8_t *base=arena.data()+Inflater::WINDOW, *o=base; size_t hist=0, tot=0;
      while(true){ bool k2=inf->run(o, base+cap, base-hist); if(!k2) break; size_t n=o-base; tot+=n; size_t t2=hist+n; size_t keep=t2<Inflater::WINDOW?t2:Inflater::WINDOW; memmove(base-keep,o-keep,keep); hist=keep; o=base; if(inf->state==Inflater::DONE) break; }
      if(tot!=data.size()) printf("SIZE MISMATCH\n"); delete inf; }
    double dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
    printf("level %d: ratio %.2f mine %.0f MB/s ok=%d eq=%d", level,(double)data.size()/comp.size(), data.size()/dt/1e6, ok, out==data);
    std::vector<uint8_t> zo; t0=std::chrono::steady_clock::now();
    { z_stream zs; memset(&zs,0,sizeof zs); inflateInit2(&zs,-15); std::vector<uint8_t> buf(4<<20); zs.next_in=(Bytef*)comp.data(); zs.avail_in=comp.size(); int rc; do{ zs.next_out=buf.data(); zs.avail_out=buf.size(); rc=inflate(&zs,Z_NO_FLUSH);}while(rc==Z_OK); inflateEnd(&zs);} dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
    printf("  zlib %.0f MB/s\n", data.size()/dt/1e6);
  }
  return nbad!=0;
}
