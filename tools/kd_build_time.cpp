// Host build time of the reference kd-tree (include/super4pcs/accelerators/kdtree.h:355-370, 522-641: midpoint split on the widest axis, <= 64 points per
// leaf, in-place partition) on 20 000 / 200 000 random surface points -- what closing divergence Q11 (the kd-tree visiting order on exact float distance ties)
// would add to every stocs_ctx_set_scene.  g++ -O3 tools/kd_build_time.cpp -o /tmp/kd && /tmp/kd  (profiles/r05_q11_cost.json)
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
struct V3 { float x,y,z; float operator[](unsigned d) const { return d==0?x:(d==1?y:z);} };
struct KdNode { float splitValue; unsigned firstChildId, dim, leaf, start, size; };
struct Kd {
  std::vector<V3> pts; std::vector<int> idx; std::vector<KdNode> nodes;
  unsigned split(int start,int end,unsigned dim,float sv){ int l(start), r(end-1); for(;l<r;++l,--r){ while(l<end && pts[l][dim]<sv) l++; while(r>=start && pts[r][dim]>=sv) r--; if(l>r) break; std::swap(pts[l],pts[r]); std::swap(idx[l],idx[r]);} return (pts[l][dim]<sv? l+1:l);} 
  void create(unsigned nodeId,unsigned start,unsigned end,unsigned level){ float mn[3]={1e30f,1e30f,1e30f}, mx[3]={-1e30f,-1e30f,-1e30f}; for(unsigned i=start;i<end;++i){ for(int k=0;k<3;++k){ float v=pts[i][k]; if(v<mn[k])mn[k]=v; if(v>mx[k])mx[k]=v; } } float dg[3]; for(int k=0;k<3;++k) dg[k]=0.5f*(mx[k]-mn[k]); unsigned dim=0; float b=dg[0]; if(dg[1]>b){b=dg[1];dim=1;} if(dg[2]>b){b=dg[2];dim=2;} nodes[nodeId].dim=dim; nodes[nodeId].splitValue=0.5f*(mx[dim]+mn[dim]); unsigned mid=split(start,end,dim,nodes[nodeId].splitValue); nodes[nodeId].firstChildId=(unsigned)nodes.size(); KdNode n; memset(&n,0,sizeof(n)); nodes.push_back(n); nodes.push_back(n);
    { unsigned c=nodes[nodeId].firstChildId; if(mid-start<=64||level>=32){nodes[c].leaf=1;nodes[c].start=start;nodes[c].size=mid-start;} else {nodes[c].leaf=0; create(c,start,mid,level+1);} }
    { unsigned c=nodes[nodeId].firstChildId+1; if(end-mid<=64||level>=32){nodes[c].leaf=1;nodes[c].start=mid;nodes[c].size=end-mid;} else {nodes[c].leaf=0; create(c,mid,end,level+1);} } }
};
int main(){ for(int n: {20000, 200000}){ std::mt19937 g(1); std::uniform_real_distribution<float> u(0,0.5f); Kd k; for(int i=0;i<n;++i){ k.pts.push_back({u(g),u(g),0.02f*u(g)}); k.idx.push_back(i);} auto t0=std::chrono::high_resolution_clock::now(); k.nodes.reserve(4*n/64+16); KdNode r; memset(&r,0,sizeof(r)); k.nodes.push_back(r); k.create(0,0,n,1); auto t1=std::chrono::high_resolution_clock::now(); printf("n=%d build %.3f ms nodes %zu\n", n, std::chrono::duration<double,std::milli>(t1-t0).count(), k.nodes.size()); } }
