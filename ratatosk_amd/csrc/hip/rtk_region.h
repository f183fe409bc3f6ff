// placeholder: region stage lands next
#ifndef RTK_REGION_H
#define RTK_REGION_H
struct RegionBatch { int dummy; };
#endif
