#include <cstdio>
#include "/root/repo/feastkit.jl_amd/csrc/fh_ingest.hpp"
template<class T> std::vector<T> rd(const char* p){ FILE* f=fopen(p,"rb"); fseek(f,0,SEEK_END); long n=ftell(f); fseek(f,0,SEEK_SET); std::vector<T> v(n/sizeof(T)); if(fread(v.data(),1,n,f)){}; fclose(f); return v; }
int main(int argc, char** argv){ auto ia=rd<int64_t>("/tmp/ia.bin"), ja=rd<int64_t>("/tmp/ja.bin"); auto va=rd<double>("/tmp/va.bin");
 int64_t N=ia.size()-1; fh_prepared<double> P; std::string err; int reorder = argc>1 ? atoi(argv[1]) : 1;
 int rc=fh_prepare_csr<double>(N,0,0,(int64_t)ja.size(),ia.data(),ja.data(),va.data(),0,nullptr,nullptr,nullptr,reorder,128,160,true,P,err);
 if(rc){printf("err %s\n",err.c_str());return 1;}
 FILE* f=fopen("/tmp/sim/rp.bin","wb"); fwrite(P.rowptr.data(),4,P.rowptr.size(),f); fclose(f);
 f=fopen("/tmp/sim/col.bin","wb"); fwrite(P.col.data(),4,P.col.size(),f); fclose(f);
 printf("N %ld nnz %zu perm %zu\n",(long)N,P.col.size(),P.perm.size()); }
