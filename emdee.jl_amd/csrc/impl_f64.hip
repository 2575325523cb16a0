// impl_f64.hip -- fp64 instantiation (north-star precision) of every kernel and handle object.
#include "dd.hpp"

namespace emdee {
template struct Factory<double>;
}
