// hooks.hpp -- call-back hooks (I/O, bitmap) of the tag API; identifiers and layout of the reference's
// interface/hooks.hpp:116-173: an entry point taking (hook, tag list) plus a client data pointer.
#ifndef MIJ_INTERFACE_HOOKS_HPP
#define MIJ_INTERFACE_HOOKS_HPP
#include "jpgtypes.hpp"
#include "tagitem.hpp"

struct JPG_EXPORT JPG_Hook {
  typedef JPG_LONG(LongHookFunction)(struct JPG_Hook *, struct JPG_TagItem *tag);
  typedef JPG_APTR(APtrHookFunction)(struct JPG_Hook *, struct JPG_TagItem *tag);
  union HookCallOut {
    JPG_LONG (*hk_pLongEntry)(struct JPG_Hook *, struct JPG_TagItem *tag);
    JPG_APTR (*hk_pAPtrEntry)(struct JPG_Hook *, struct JPG_TagItem *tag);
    HookCallOut(LongHookFunction *h) : hk_pLongEntry(h) {}
    HookCallOut(APtrHookFunction *h) : hk_pAPtrEntry(h) {}
    HookCallOut() : hk_pLongEntry(0) {}
  } hk_Entry, hk_SubEntry;
  JPG_APTR hk_pData; // for the client

  JPG_Hook(LongHookFunction *hook = 0, JPG_APTR data = 0) : hk_Entry(hook), hk_pData(data) {}
  JPG_Hook(APtrHookFunction *hook, JPG_APTR data = 0) : hk_Entry(hook), hk_pData(data) {}
  JPG_LONG CallLong(struct JPG_TagItem *tag) { return (*hk_Entry.hk_pLongEntry)(this, tag); }
  JPG_APTR CallAPtr(struct JPG_TagItem *tag) { return (*hk_Entry.hk_pAPtrEntry)(this, tag); }
};
#endif
