// Stand-in for the two boost::serialization names the reference's DBoW2 headers mention (TEST INFRASTRUCTURE): the
// serialize() member templates that use them are never instantiated here.
#pragma once
namespace boost { namespace serialization {
class access;
template <class Base, class Derived> Base& base_object(Derived& d) { return static_cast<Base&>(d); }
} }
