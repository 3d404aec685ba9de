#include "serialization.hpp"
